"""screencounter_amd -- MI355X-native barcode counting engine behind screenCounter's hot path.

Package contents (only what the path needs):
  csrc/      HIP kernels (gfx950) + host runtime + the C ABI of include/scg.h -> libscg.so
  _lib.py    ctypes binding of libscg.so (fails loudly when the library is missing)
  engine.py  Plan objects over device-resident read batches
  api.py     mirror of the reference's Rcpp-level entry points for this path (same names, arguments, results)
  synth.py   synthetic workloads of SURVEY.md section 8(d) (device-side generator)
  parallel.py  read-sharded multi-GPU driver (torch.distributed over RCCL)
"""
from ._lib import ScgError, load, prepare_pool  # noqa: F401
from .api import (  # noqa: F401
    count_single_barcodes, count_combo_barcodes_single, count_dual_barcodes, count_combo_barcodes_paired,
    count_single_barcodes_files, count_combo_barcodes_single_files, count_dual_barcodes_files, count_dual_barcodes_single_end,
    count_random_barcodes, match_barcodes, parse_fastq,
)
from .engine import Plan, combo_compact, upload_reads  # noqa: F401

__version__ = "0.3.0"
