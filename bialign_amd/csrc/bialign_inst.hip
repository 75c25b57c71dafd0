// bialign_inst.hip -- one (max_shift, kind) slice of the kernel launchers; see bialign_host.hpp.
// Compiled with -DBIALIGN_TU_S=<0..5> -DBIALIGN_TU_KIND=<0|1>.
#include "bialign_host.hpp"

namespace bialign {
#if BIALIGN_TU_KIND == 0
BIALIGN_INST_KIND0(BIALIGN_TU_S, )
#else
BIALIGN_INST_KIND1(BIALIGN_TU_S, )
#endif
}  // namespace bialign
