// scg_launch.h -- launchers exported by scg_kernels.hip to the host runtime.
#ifndef SCG_LAUNCH_H
#define SCG_LAUNCH_H

#include <hip/hip_runtime_api.h>

#include "../../include/scg.h"
#include "scg_common.h"

namespace scg {

// Test hook: SCG_FORCE_GENERAL=1 (any value but empty / 0) runs the byte-wise engine alone; read at every launch.
bool force_general();

hipError_t launch_single(const ScgSingleParams& P, int tmpl_len, const ScgReads& R, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream);
// countRandomBarcodes: d_hits[i] = (position << 1) | reverse of read i's template hit, or -1
hipError_t launch_random(const ScgSingleParams& P, int tmpl_len, const ScgReads& R, int64_t n, int32_t* d_hits, int32_t* flag, hipStream_t stream);
hipError_t launch_combo(const ScgComboParams& P, int tmpl_len, const ScgReads& R, int64_t n, const ScgCounters& cells, int32_t* flag, hipStream_t stream);
hipError_t launch_dual(const ScgDualParams& P, int tmpl_len, const ScgReads& R1, const ScgReads& R2, int64_t n, const ScgCounters& counts, int32_t* flag, hipStream_t stream);
hipError_t launch_fold(int32_t* replicas, int replica_shift, int64_t n, int32_t* counters, hipStream_t stream);
// hot[2][SCG_HOT_SLOTS] -> pair_of_counters[0], [1] (and clears the slots); ScgCounters::hot
hipError_t launch_hot_fold(int32_t* hot, int32_t* pair_of_counters, hipStream_t stream);
// counters[unit_index[r]] += 1 for every r with unit_index[r] >= 0, through LDS histograms (ScgCounters::unit_index)
hipError_t launch_tally(const int32_t* unit_index, int64_t n, int32_t* counters, int64_t n_counters, hipStream_t stream);
// Combination streams (ScgCounters::unit_pair) -> runs of (distinct key, count): scg_sparse.hip.  d_sorted, d_unique: n keys
// each; d_counts: n; d_runs: 1; scratch of sort_rle_scratch_bytes(n).
size_t sort_rle_scratch_bytes(size_t n);
hipError_t launch_sort_rle(const uint64_t* d_keys, uint64_t* d_sorted, size_t n, uint64_t* d_unique, uint32_t* d_counts, uint32_t* d_runs,
                           void* d_scratch, size_t scratch_bytes, hipStream_t stream);
hipError_t launch_match(const ScgIndex& tab, const uint8_t* d_seqs, int32_t n, int cap, int reverse,
                        int32_t* d_index, int32_t* d_mm, hipStream_t stream);
hipError_t launch_synth(const scg_synth_spec& S, char* d_out, int64_t n, hipStream_t stream);

} // namespace scg

#endif
