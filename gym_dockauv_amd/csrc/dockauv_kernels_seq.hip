// float instantiations of the resident step-sequence kernel (dockauv_step_sequence's fast path)
#define DOCKAUV_INSTANTIATE_SEQ 1
#include "dockauv_step.hip.inc"
