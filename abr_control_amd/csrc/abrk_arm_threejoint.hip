// Compile-time specialised kernels of the built-in "threejoint" arm (both arithmetic types).
#include "abrk_kernels.h"
namespace abrk {
const ArmOps* ops_threejoint() { return OpsFor<StaticArm<Tab_threejoint>, StaticArm<Tab_threejoint>>::ops(); }
}  // namespace abrk
