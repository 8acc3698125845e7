// drone_kernel_k.hip -- the drone_kernel instantiations of ONE k_closest value (csrc/Makefile compiles this file eight
// times, -DDRONESIM_K=1..8, side by side: 7 geometry / far x 5 mode / episode-layer combinations each); dronesim.hip
// dispatches to `dronesim_launch_k<K>`.
#include "drone_kernel.hpp"

#define DRONESIM_LAUNCH_K2(n) dronesim_launch_k##n
#define DRONESIM_LAUNCH_K(n) DRONESIM_LAUNCH_K2(n)
extern "C" __attribute__((visibility("hidden")))
int DRONESIM_LAUNCH_K(DRONESIM_K)(int mode, int far, const void *a, const void *g, void *s)
{
    return (int)launch_k<DRONESIM_K>(mode, far != 0, *static_cast<const KArgs *>(a), *static_cast<const Geometry *>(g),
                                     static_cast<hipStream_t>(s));
}
