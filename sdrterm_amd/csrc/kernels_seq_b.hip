// Second translation unit of the per-sample chain kernel: its decimation-4 instantiations (see launch_seq_every in kernels_seq.hip).
#define SDRT_SEQ_PART_B 1
#include "kernels_seq.hip"
