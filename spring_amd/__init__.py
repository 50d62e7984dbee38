"""spring_amd -- MI355X-native replacement for SPRING's read-reordering stage
(reference src/reorder.h + bitset_util.{h,cpp}).  Python is only the host-side
mirror of the stage interface; all compute is in lib/libspring_reorder_hip.so
(hand-written HIP for gfx950).  There is no CPU fallback: importing works
anywhere, running requires the built library and a GPU."""
from .reorder import (CompressionParams, ReorderError, ReorderOpts, ReorderStage, call_reorder, reorder_dna,  # noqa: F401
                      synth_dna_host, synth_genome_host, SYNTH_GENOMIC, SYNTH_PAIRED, SYNTH_REPEATS)
