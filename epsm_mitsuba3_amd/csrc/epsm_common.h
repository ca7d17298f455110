// epsm_common.h -- shared by the translation units of libepsm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/epsm.h"

namespace epsm_host {

char *err_buf();                       // thread-local, 512 bytes
int fail(int code, const char *what, const char *detail = "");
int hip_fail(const char *what, hipError_t e);

}  // namespace epsm_host
