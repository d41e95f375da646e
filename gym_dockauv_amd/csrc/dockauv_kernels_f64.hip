// double instantiations of the step kernel (validation path)
#define DOCKAUV_INSTANTIATE_F64 1
#include "dockauv_step.hip.inc"
