// Kernel instantiations for the table-driven LTV dynamics with n = 12, m = 4, fp64.
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_ltv12x4_f64(bool diag) { return diag ? Launch<double, Ltv12x4Model, true>::table() : nullptr; }
} // namespace ilqr
