// passes_simple.hip -- the ray kernels of passes.hip once more, compiled for "simple" frames: every texture of the frame has power-of-two
// sizes and every ray-traced instance is shadow-opaque (rule O2), so the non-power-of-two texel addressing and the shadow any-hit program
// (rt64_shader.cpp:594-663) are left out.  Same arithmetic on the paths that remain: results are bit-identical to the general kernels
// (tests/test_gpu_features.py compares the two).  View::update decides per frame (FrameParams::simpleKernels).
#define RT_ASSUME_SIMPLE 1
#include "passes.hip"
