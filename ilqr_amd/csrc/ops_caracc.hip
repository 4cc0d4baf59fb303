// Kernel instantiations for the CarAcc dynamics (fp64 and fp32, diagonal and dense cost weights).
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_caracc(int dtype, bool diag) { return ops_for<CarAccModel>(dtype, diag); }
} // namespace ilqr
