// bulk_kernel_wide.hip — the search kernel for the InterX checker and automata of more than 64 trims (the realistic automaton,
// choose_trims.m:85-131: 71 trims): bulk_search.hpp with the number of successor-mask words read at run time.
#include "bulk_search.hpp"

PDMPC_BULK_KERNEL(pdmpc_bulk_kernel_wide, pdmpc_launch_bulk_wide, 0, PDMPC_CHECK_INTERX, PDMPC_MAX_WAVES)
