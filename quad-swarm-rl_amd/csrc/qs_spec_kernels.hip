// qs_spec_kernels.hip - translation unit of a config-specialised code object (see qs_kernels.h):
//   hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -DQS_SPEC_FILE='"<header>"' qs_spec_kernels.hip -o qs_<key>.hsaco
// exports qs_spec_step / qs_spec_rollout / qs_spec_reset for exactly the configuration the header describes.
// Built on demand by qs_create() (or ahead of time by qs_spec_build()) and cached next to the library.
#ifndef QS_SPEC_FILE
#error "compile with -DQS_SPEC_FILE=\"<header written by qs_spec_header()>\""
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include QS_SPEC_FILE
#define QS_SPEC 1
#include "qs_kernels.h"
