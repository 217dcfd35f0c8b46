// Compile-time specialised kernels of the built-in "onejoint" arm (both arithmetic types).
#include "abrk_kernels.h"
namespace abrk {
const ArmOps* ops_onejoint() { return OpsFor<StaticArm<Tab_onejoint>, StaticArm<Tab_onejoint>>::ops(); }
}  // namespace abrk
