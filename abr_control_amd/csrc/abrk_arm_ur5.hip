// Compile-time specialised kernels of the built-in "ur5" arm (both arithmetic types).
#include "abrk_kernels.h"
namespace abrk {
const ArmOps* ops_ur5() { return OpsFor<StaticArm<Tab_ur5>, StaticArm<Tab_ur5>>::ops(); }
}  // namespace abrk
