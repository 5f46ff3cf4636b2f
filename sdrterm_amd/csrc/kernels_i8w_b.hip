// Second translation unit of the wide-column kernel: the 8-bit encodings (see the end of kernels_i8w.hip).
#define SDRT_I8W_BYTES 1
#include "kernels_i8w.hip"
