// bulk_kernel_sat.hip — the search kernel for the separating-axis checker (are_constraints_satisfied_sat.m, intersect_sat.m,
// intersect_lanelet_boundary.m: the circle scenario and every configuration with convex obstacles only), any automaton.
#include "bulk_search.hpp"

PDMPC_BULK_KERNEL(pdmpc_bulk_kernel_sat, pdmpc_launch_bulk_sat, 0, PDMPC_CHECK_SAT, PDMPC_MAX_WAVES_SAT)
