// Compile-time specialised kernels of the built-in "twojoint" arm (both arithmetic types).
#include "abrk_kernels.h"
namespace abrk {
const ArmOps* ops_twojoint() { return OpsFor<StaticArm<Tab_twojoint>, StaticArm<Tab_twojoint>>::ops(); }
}  // namespace abrk
