// Kernel instantiations of the sequential-QP solver (Car dynamics, fp64).
#include "host.hpp"
#include "sqp.hpp"
namespace ilqr {
const SqpOps *sqp_ops(int model, int dtype) {
    if (model == ILQR_MODEL_CAR && dtype == ILQR_F64) return SqpLaunch<double, CarModel>::table();
    return nullptr;
}
} // namespace ilqr
