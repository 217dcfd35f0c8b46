// Joint-count dispatch for user arms.
#include "abrk_rt.h"
namespace abrk {
const ArmOps* ops_rt(int n) {
  switch (n) {
    case 1: return ops_rt1();
    case 2: return ops_rt2();
    case 3: return ops_rt3();
    case 4: return ops_rt4();
    case 5: return ops_rt5();
    case 6: return ops_rt6();
    case 7: return ops_rt7();
  }
  return nullptr;
}
size_t rt_table_size(int n, int dtype) {
  switch (n) {
    case 1: return rt_size1(dtype);
    case 2: return rt_size2(dtype);
    case 3: return rt_size3(dtype);
    case 4: return rt_size4(dtype);
    case 5: return rt_size5(dtype);
    case 6: return rt_size6(dtype);
    case 7: return rt_size7(dtype);
  }
  return 0;
}
void rt_table_fill(int n, int dtype, const abrk_arm_desc* d, void* dst) {
  switch (n) {
    case 1: rt_fill1(dtype, d, dst); break;
    case 2: rt_fill2(dtype, d, dst); break;
    case 3: rt_fill3(dtype, d, dst); break;
    case 4: rt_fill4(dtype, d, dst); break;
    case 5: rt_fill5(dtype, d, dst); break;
    case 6: rt_fill6(dtype, d, dst); break;
    case 7: rt_fill7(dtype, d, dst); break;
  }
}
}  // namespace abrk
