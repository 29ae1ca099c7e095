// bulk_kernel.hip — the product's search kernel for the InterX checker and automata of up to 64 trims (every BASELINE road-network
// configuration): bulk_search.hpp instantiated with one successor-mask word.
#include "bulk_search.hpp"

PDMPC_BULK_KERNEL(pdmpc_bulk_kernel, pdmpc_launch_bulk, 1, PDMPC_CHECK_INTERX, PDMPC_MAX_WAVES)
