/* orbx_synth.h -- synthetic input frames for the tests and bench.py (liborbx_synth.so).
 *
 * NOT part of the product: liborbx.so does not contain it.  Datasets are absent and there is no network, so every benchmark / parity
 * input is rendered by this host-only, integer-only, deterministic generator. */
#ifndef ORBX_SYNTH_H
#define ORBX_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_SYNTH_LOW_TEXTURE 1  /* few shapes, +-2 noise: hits the minThFAST fallback */
#define ORBX_SYNTH_STEREO_RIGHT 2 /* right view of the same scene, per-shape disparity   */
int orbx_synth_frame(uint64_t seed, int width, int height, int stride, int flags, uint8_t *dst);
/* View `view` of scene `seed`: all shapes translated by (dx,dy) pixels, fresh noise per view
 * (consecutive views share most corners: frame-to-frame matching has something to match). */
int orbx_synth_frame_ex(uint64_t seed, int view, int dx, int dy, int width, int height, int stride,
                        int flags, uint8_t *dst);


#ifdef __cplusplus
}
#endif

#endif
