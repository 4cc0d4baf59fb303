// Kernel instantiations for the table-driven LTV dynamics with n = 12, m = 4, fp32.
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_ltv12x4_f32(bool diag) { return diag ? Launch<float, Ltv12x4Model, true>::table() : nullptr; }
} // namespace ilqr
