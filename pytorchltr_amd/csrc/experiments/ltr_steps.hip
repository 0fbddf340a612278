// ltr_steps.hip -- translation unit of the persistent multi-batch training kernel (ltr_steps.inc) of libltr_hip.so;
// compiled next to ltr_kernels.hip, ltr_linear.hip and ltr_mlp.hip (pytorchltr_amd/build.py).
#include <deque>
#include <map>
#include <mutex>
#include <string.h>

#include "ltr_common.inc"
#include "ltr_steps.inc"
