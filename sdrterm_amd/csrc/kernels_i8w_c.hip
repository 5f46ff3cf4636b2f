// third translation unit of the wide-column kernel: the unsigned 16-bit encoding (see kernels_i8w.hip)
#define SDRT_I8W_U16
#include "kernels_i8w.hip"
