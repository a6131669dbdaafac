// The content stamp of the build (ptlflow_amd/_build.py passes -DPFK_SOURCE_HASH): its own tiny translation unit, so that a
// change anywhere in csrc/ recompiles this file and the changed one, not the big kernel files.
#include "pfk.h"

#ifndef PFK_SOURCE_HASH
#define PFK_SOURCE_HASH "unstamped"
#endif

// The marker in front lets the build script read the stamp out of the FILE (ptlflow_amd/_build.py embedded_hash) without
// dlopen-ing a library it may be about to relink.
static const char pfk_stamp_record[] = "PFK_SOURCE_HASH=" PFK_SOURCE_HASH;

extern "C" const char* pfk_source_hash(void) { return pfk_stamp_record + 16; }
