/* oracle_svgf.c -- SVGF denoiser of the CPU oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).  Placeholder: pass-through. */
#include <string.h>
#include "oracle_internal.h"

void osvgf_filter(OScene *s, const OFrameParams *p, int cur) {
    size_t n = (size_t)p->width * (size_t)p->height;
    memcpy(s->filteredIndirect[1], s->indirectLight[cur], n * 4 * sizeof(float));
}
