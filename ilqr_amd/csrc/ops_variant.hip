// One (model, element type, diagonal / dense cost weights) set of kernel instantiations, chosen by
// the macros OPS_MODEL (CarModel ..), OPS_T (double, float), OPS_DIAG (true, false) and OPS_FN (the
// accessor's name).  The Makefile compiles this file once per combination, so that the variants of a
// model build side by side instead of one after the other in a single translation unit.
#include "host.hpp"
namespace ilqr {
const ModelOps *OPS_FN() { return Launch<OPS_T, OPS_MODEL, OPS_DIAG>::table(); }
} // namespace ilqr
