// The content stamp of the build (ptlflow_amd/_build.py passes -DPFK_SOURCE_HASH): its own tiny translation unit, so that a
// change anywhere in csrc/ recompiles this file and the changed one, not the big kernel files.
#include "pfk.h"

#ifndef PFK_SOURCE_HASH
#define PFK_SOURCE_HASH "unstamped"
#endif

extern "C" const char* pfk_source_hash(void) { return PFK_SOURCE_HASH; }
