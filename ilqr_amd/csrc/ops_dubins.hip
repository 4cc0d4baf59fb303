// Kernel instantiations for the Dubins dynamics (fp64 and fp32, diagonal and dense cost weights).
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_dubins(int dtype, bool diag) { return ops_for<DubinsModel>(dtype, diag); }
} // namespace ilqr
