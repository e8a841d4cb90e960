// window_inst.hip -- instantiation unit of window_kernel (locate_fast.hpp).  Compiled 4
// times with -DATR_WIN_GROUP=g; unit g holds MT = ROW_GRAN * (4g+1 .. 4g+4).
#include "locate_fast.hpp"

#ifndef ATR_WIN_GROUP
#error "compile with -DATR_WIN_GROUP=<0..3>"
#endif
#define ATR_CAT2(a, b) a##b
#define ATR_CAT(a, b) ATR_CAT2(a, b)

namespace atr {
constexpr int W0 = ATR_WIN_GROUP * 4;
window_launcher ATR_CAT(window_group_, ATR_WIN_GROUP)(int i) {
    switch (i) {
        case 0: return &launch_window_mt<ROW_GRAN *(W0 + 1)>;
        case 1: return &launch_window_mt<ROW_GRAN *(W0 + 2)>;
        case 2: return &launch_window_mt<ROW_GRAN *(W0 + 3)>;
        default: return &launch_window_mt<ROW_GRAN *(W0 + 4)>;
    }
}
}  // namespace atr
