// tests/fake_rcpp/Rcpp.h -- a TYPE-LEVEL stand-in for the handful of Rcpp / R API names shim/scg_shim.cpp uses,
// so that the shim can be syntax- and type-checked in an image without R (tests/test_shim.py runs
// `g++ -fsyntax-only`).  Nothing here is functional and nothing is ever linked; with the real Rcpp the shim
// compiles unchanged.  Test infrastructure only.
#ifndef FAKE_RCPP_H
#define FAKE_RCPP_H

#include <cstddef>
#include <string>

typedef std::ptrdiff_t R_xlen_t;
struct SEXPREC;
typedef SEXPREC* SEXP;
const char* CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
extern int NA_INTEGER;

namespace Rcpp {

[[noreturn]] void stop(const char* msg);
[[noreturn]] void stop(const std::string& msg);

struct ListProxy { operator SEXP() const; template<class T> ListProxy& operator=(const T&); };

struct CharacterVector {
    struct Proxy { Proxy& operator=(const std::string&); Proxy& operator=(const char*); };
    CharacterVector();
    explicit CharacterVector(R_xlen_t n);
    CharacterVector(SEXP);
    CharacterVector(const ListProxy&);
    R_xlen_t size() const;
    Proxy operator[](R_xlen_t);
    operator SEXP() const;
};

struct IntegerVector {
    IntegerVector();
    explicit IntegerVector(R_xlen_t n);
    template<class It> IntegerVector(It first, It last);
    int* begin();
    R_xlen_t size() const;
    int& operator[](R_xlen_t);
    static IntegerVector create(int);
    operator SEXP() const;
};

struct IntegerMatrix {
    IntegerMatrix(int nrow, R_xlen_t ncol);
    int* begin();
    int* column_begin(R_xlen_t);
    operator SEXP() const;
};

struct List {
    List();
    explicit List(R_xlen_t n);
    R_xlen_t size() const;
    ListProxy operator[](R_xlen_t) const;
    template<class... Args> static List create(const Args&...);
    operator SEXP() const;
};

} // namespace Rcpp

#endif
