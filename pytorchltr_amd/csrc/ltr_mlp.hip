// ltr_mlp.hip -- translation unit of the fused MLP scorer + loss kernels (ltr_mlp.inc, ltr_mlp2.inc)
// and the stand-alone Linear scorer layer (ltr_scorer.inc, which shares the MLP's reduction kernel).
#include "ltr_common.inc"
#include "ltr_mlp.inc"
#include "ltr_scorer.inc"
