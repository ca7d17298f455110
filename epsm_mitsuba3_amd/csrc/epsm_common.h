// epsm_common.h -- shared by the translation units of libepsm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/epsm.h"
#include "../../include/epsm_trace.h"

namespace epsm_host {

char *err_buf();                       // thread-local, 512 bytes
int fail(int code, const char *what, const char *detail = "");
int hip_fail(const char *what, hipError_t e);

// What every tracer entry point checks of the scene's emitter / environment / texture tables before a kernel may touch them
// (include/epsm_trace.h, EpsmEnvironment): NULL = fine, otherwise what is wrong.
inline const char *scene_tables_invalid(const EpsmScene *s) {
    if (s->n_emitters < 0 || (s->n_emitters > 0 && !s->emitters)) return "NULL emitters";
    const EpsmEnvironment &e = s->env;
    if (e.kind != EPSM_ENV_NONE && e.kind != EPSM_ENV_CONSTANT && e.kind != EPSM_ENV_ENVMAP) return "env.kind is not an EPSM_ENV_* value";
    if (e.kind != EPSM_ENV_NONE && (e.emitter < 0 || e.emitter >= s->n_emitters)) return "env.emitter is not an index into emitters";
    if (e.kind == EPSM_ENV_ENVMAP && (!e.texels || !e.row_cdf || !e.col_cdf || !e.cell_pdf || e.width < 2 || e.height < 2))
        return "envmap environment needs texels, row_cdf, col_cdf, cell_pdf and width, height >= 2";
    if (s->n_textures < 0 || (s->n_textures > 0 && !s->textures)) return "NULL textures";
    return nullptr;
}

}  // namespace epsm_host
