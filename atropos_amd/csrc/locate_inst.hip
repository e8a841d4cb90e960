// locate_inst.hip -- explicit instantiation unit of locate_kernel.  Compiled
// LOCATE_GROUPS times with -DATR_INST_GROUP=g; unit g holds the column sizes
// MT = ROW_GRAN * (g * LOCATE_PER_GROUP + 1 ... g * LOCATE_PER_GROUP + LOCATE_PER_GROUP).
#include "locate_kernel.hpp"

#ifndef ATR_INST_GROUP
#error "compile with -DATR_INST_GROUP=<0..7>"
#endif

#define ATR_CAT2(a, b) a##b
#define ATR_CAT(a, b) ATR_CAT2(a, b)

namespace atr {
constexpr int G0 = ATR_INST_GROUP * LOCATE_PER_GROUP;
static_assert(LOCATE_PER_GROUP == 4, "switch below lists four sizes");
// Host-only accessor (a global table of host function pointers would also be emitted
// for the device side).
locate_launcher ATR_CAT(locate_group_, ATR_INST_GROUP)(int i) {
    switch (i) {
        case 0: return &launch_locate_mt<ROW_GRAN *(G0 + 1)>;
        case 1: return &launch_locate_mt<ROW_GRAN *(G0 + 2)>;
        case 2: return &launch_locate_mt<ROW_GRAN *(G0 + 3)>;
        default: return &launch_locate_mt<ROW_GRAN *(G0 + 4)>;
    }
}
}  // namespace atr
