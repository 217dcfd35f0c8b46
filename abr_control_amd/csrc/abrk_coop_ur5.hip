// The wave-cooperative OSC kernels (K = 4, 8, 16 lanes per arm) of the built-in "ur5" arm - abrk_coop.h.
#include "abrk_coop.h"
namespace abrk {
hipError_t launch_osc_coop_ur5(const LaunchArgs& la, const CoopArgs& a) { return launch_osc_coop<StaticArm<Tab_ur5>>(la, a); }
}  // namespace abrk
