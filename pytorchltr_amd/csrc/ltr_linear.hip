// ltr_linear.hip -- translation unit of the fused Linear(F,1) scorer + loss kernels
// (ltr_linear.inc, ltr_cluster.inc) of libltr_hip.so; compiled next to ltr_kernels.hip and
// ltr_mlp.hip (pytorchltr_amd/build.py), so the three build in parallel.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string.h>
#include <thread>

#include "ltr_common.inc"
#include "ltr_linear.inc"
