// pgx_sketch_fast.hip -- closed-form wavefront-per-read minimizer sketch (placeholder: everything is routed to
// the literal kernel until the wave kernel lands).
#include "pgx_internal.h"
namespace pgx {
bool sketch_wave_eligible(const ReadDesc &, int, int) { return false; }
void launch_sketch_wave(const pgx_seqdb *, const ReadDesc *, const uint32_t *, uint32_t, int, int, int, uint32_t *,
                        const uint64_t *, pgx_mm128 *, uint32_t *) {}
}  // namespace pgx
