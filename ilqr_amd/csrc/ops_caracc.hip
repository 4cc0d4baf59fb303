// Launch tables of the caracc dynamics (fp64 and fp32, diagonal and dense cost weights); the kernels
// themselves are instantiated by ops_variant.hip, one translation unit per combination.
#include "host.hpp"
namespace ilqr {
const ModelOps *ops_caracc_f64_diag();
const ModelOps *ops_caracc_f64_dense();
const ModelOps *ops_caracc_f32_diag();
const ModelOps *ops_caracc_f32_dense();
const ModelOps *ops_caracc(int dtype, bool diag) {
    if (dtype == ILQR_F64) return diag ? ops_caracc_f64_diag() : ops_caracc_f64_dense();
    if (dtype == ILQR_F32) return diag ? ops_caracc_f32_diag() : ops_caracc_f32_dense();
    return nullptr;
}
} // namespace ilqr
