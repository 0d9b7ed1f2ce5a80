// Load-time identity of libfvqa_hip.so: the ABI version (bumped on any signature change of include/fvqa.h) and the hash of
// the kernel sources the library was built from (fvqa/build.py passes -DFVQA_SOURCE_HASH; fvqa/_lib.py compares it with the
// sources it finds next to the library and refuses a stale binary).
#include "../../include/fvqa.h"

#ifndef FVQA_SOURCE_HASH
#define FVQA_SOURCE_HASH "unknown"
#endif

extern "C" int fvqa_version(void) { return 16; }
extern "C" const char* fvqa_source_hash(void) { return FVQA_SOURCE_HASH; }
