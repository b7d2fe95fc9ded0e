// kernels_net_deep.hip -- the layer-at-a-time MLP kernels (kernels_net.hip) for three and four hidden layers: a translation unit of its own (build time)
#define MON_NET_PART 2
#include "kernels_net.hip"
