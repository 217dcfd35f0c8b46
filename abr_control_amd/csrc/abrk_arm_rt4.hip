// Generic kernels for user arms with 4 joint(s) (table passed at launch), both dtypes.
#include "abrk_rt.h"
namespace abrk {
ABRK_RT_DEFINE(4)
}  // namespace abrk
