// Kernel instantiations for the Car dynamics (fp64 and fp32, diagonal and dense cost weights).
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_car(int dtype, bool diag) { return ops_for<CarModel>(dtype, diag); }
} // namespace ilqr
