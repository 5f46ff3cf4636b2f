// fourth translation unit of the wide-column kernel: float32 records (see kernels_i8w.hip)
#define SDRT_I8W_F32
#include "kernels_i8w.hip"
