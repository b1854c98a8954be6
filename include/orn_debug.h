/* orn_debug.h -- probe-only entry points of liborn.so (tools/probes).  NOT part of the drop-in boundary: nothing here has a
 * reference counterpart, and the flags make results WRONG (they exist to time parts of a kernel in isolation). */
#ifndef ORN_DEBUG_H_
#define ORN_DEBUG_H_
#include "orn.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Timing-only ablation flags of the 16-bit conv / wgrad kernels; effective only in a library built with -DORN_CONV_ABLATE
 * (the product build compiles the switches out).  0 restores normal operation. */
ORN_API void orn_debug_set(int flags);
#ifdef ORN_CONV_STAMP
/* Diagnostic build -DORN_CONV_STAMP only: buffer of 128 uint64 per work-group that receives the conv kernel's phase stamps. */
ORN_API void orn_debug_set_stamps(void *buf);
#endif
#ifdef __cplusplus
}
#endif
#endif /* ORN_DEBUG_H_ */
