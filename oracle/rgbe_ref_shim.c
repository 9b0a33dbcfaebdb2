/*
 * rgbe_ref_shim.c -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).
 *
 * Drives the REFERENCE'S OWN Radiance RGBE reader / writer, Common/rgbe.c, through memory buffers so that tests can
 * compare the product's codec (cudapathtracer_amd/cpp/Rgbe.hpp, mipt_rgbe_*) with it byte for byte.  rgbe.c is plain C
 * and compiles here unmodified; `make -C oracle ref` compiles it from where it lies under /root/reference together
 * with this file into oracle/_ref/librgbe_ref.so.  No line of the reference is copied into this repository; this file
 * only calls its public functions (Common/rgbe.h:35-48) in the order DefaultScene/tracer.cpp:130-134 does:
 *     RGBE_ReadHeader(f, &w, &h, NULL);  RGBE_ReadPixels_RLE(f, image, w, h);
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rgbe.h"

/* Decodes a whole .hdr file held in memory with the reference's reader.
 *   returns 0 = accepted (pixels written if rgb_out != NULL),
 *           1 = RGBE_ReadHeader failed, 2 = RGBE_ReadPixels_RLE failed,
 *           3 = the header was accepted but names a size the harness will not hand to the reader: non-positive, or more
 *               floats than cap_floats (the reference itself would index a vector of 3*w*h floats: DefaultScene/tracer.cpp:132).
 * width / height are written whenever the header was accepted.  rgb_out == NULL: header only. */
int rgbe_ref_decode(const unsigned char* data, size_t size, float* rgb_out, size_t cap_floats, int* width, int* height) {
    int w = 0, h = 0, rc;
    FILE* f;
    if (size == 0) return 1;                       /* fgets on an empty file: read error (rgbe.c:147-148) */
    f = fmemopen((void*)data, size, "rb");
    if (!f) return 1;
    rc = RGBE_ReadHeader(f, &w, &h, NULL);
    if (rc != RGBE_RETURN_SUCCESS) { fclose(f); return 1; }
    if (width) *width = w;
    if (height) *height = h;
    if (w <= 0 || h <= 0) { fclose(f); return 3; }
    if (!rgb_out) { fclose(f); return 0; }
    if ((unsigned long long)w * (unsigned long long)h * 3ull > (unsigned long long)cap_floats) { fclose(f); return 3; }
    rc = RGBE_ReadPixels_RLE(f, rgb_out, w, h);
    fclose(f);
    return rc == RGBE_RETURN_SUCCESS ? 0 : 2;
}

/* Encodes float RGB rows with the reference's writer (RGBE_WriteHeader + RGBE_WritePixels_RLE); returns the byte count
 * (0 on failure); writes at most cap bytes to out (call with out == NULL to size the buffer). */
size_t rgbe_ref_encode(const float* rgb, int width, int height, unsigned char* out, size_t cap) {
    char* buf = NULL; size_t len = 0;
    FILE* f = open_memstream(&buf, &len);
    int ok;
    if (!f) return 0;
    ok = RGBE_WriteHeader(f, width, height, NULL) == RGBE_RETURN_SUCCESS &&
         RGBE_WritePixels_RLE(f, (float*)rgb, width, height) == RGBE_RETURN_SUCCESS;
    fclose(f);
    if (!ok) { free(buf); return 0; }
    if (out && len <= cap) memcpy(out, buf, len);
    free(buf);
    return len;
}
