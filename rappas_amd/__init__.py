"""MI355X-native phylo-kmer placement engine for RAPPAS's `-p p` hot path.

The product is the C-ABI library `librappas_place.so` (include/rappas_place.h, gfx950 HIP kernels in
rappas_amd/csrc/).  This package is the thin host-side mirror used by tests and bench.py.
"""
from . import _lib, build, synth  # noqa: F401
from ._lib import (RK_ALPHABET_AA, RK_ALPHABET_DNA, RK_AMB_MAX, RK_AMB_MEAN, RK_AMB_SKIP, RK_FLAG_AMBIGUOUS,  # noqa: F401
                   RK_FLAG_BAD_CHAR, RK_FLAG_BELOW_NSBOUND, RK_FLAG_PLACED, RK_FLAG_TOO_LONG, RK_FLAG_TOO_SHORT,
                   RK_TABLE_AUTO, RK_TABLE_DIRECT, RK_TABLE_DIRECT8, RK_TABLE_HASH, RkError)
from .placement import (PhyloKmerDB, PlacementProcess, Placements, db_image_info, host_alloc, pack_reads, save_db_image,  # noqa: F401
                        validate_db)
from .dbbuild import BuiltDB, build_db  # noqa: F401
