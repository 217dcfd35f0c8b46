// Compile-time specialised kernels of the built-in "jaco2" arm (both arithmetic types).
#include "abrk_kernels.h"
namespace abrk {
const ArmOps* ops_jaco2() { return OpsFor<StaticArm<Tab_jaco2>, StaticArm<Tab_jaco2>>::ops(); }
}  // namespace abrk
