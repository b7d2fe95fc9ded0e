// kernels_net_wide.hip -- the layer-at-a-time MLP kernels (kernels_net.hip) for 16 and 128 neurons: a translation unit of its own so that the build compiles it beside the others
#define MON_NET_PART 1
#include "kernels_net.hip"
