// lib_dist.hip -- the Z-slab sharded step (dist_rccl.h: RCCL / host transport behind the protocol of slab_protocol.h) and several GPUs
// from one process (node_local.h).
#include "lib_internal.h"

#define SDFK_LIB_DIST_TU 1
#include "dist_rccl.h"
#include "node_local.h"

