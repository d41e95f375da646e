// float instantiations of the step kernel (product path)
#define DOCKAUV_INSTANTIATE_F32 1
#include "dockauv_step.hip.inc"
