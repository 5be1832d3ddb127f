// avrf_all.hip -- single translation unit of libavrf.so (no relocatable device code needed;
// lets hipcc inline the field/curve templates into every kernel).
#include "msm.hip"
#include "vrf_batch.hip"
#include "capi.hip"
