/* pt_image_io.c — minimal PNG / BMP / TGA writers, used only when the host is built WITHOUT
 * stb_image_write.h (the reference's vendored encoder, include/stb_image_write.h, which a drop-in
 * build picks up from the reference tree with -DPT_HAVE_STB; see renderer.c and the Makefile).
 * They write the same pixels in the same formats (8-bit RGB), with the simplest legal encoding:
 * PNG uses stored (uncompressed) deflate blocks.  JPEG is not provided here.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_image_io.h"

static uint32_t crc_table[256];
static int crc_ready = 0;

static void crc_init(void) {
	for(uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for(int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		crc_table[n] = c;
	}
	crc_ready = 1;
}

static uint32_t crc_update(uint32_t c, const unsigned char *p, size_t n) {
	for(size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
	return c;
}

static void put_be32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

static int write_chunk(FILE *f, const char *tag, const unsigned char *data, uint32_t len) {
	unsigned char hdr[8];
	put_be32(hdr, len);
	memcpy(hdr + 4, tag, 4);
	uint32_t c = crc_update(0xFFFFFFFFu, hdr + 4, 4);
	if(len) c = crc_update(c, data, len);
	unsigned char tail[4];
	put_be32(tail, c ^ 0xFFFFFFFFu);
	if(fwrite(hdr, 1, 8, f) != 8) return 0;
	if(len && fwrite(data, 1, len, f) != len) return 0;
	return fwrite(tail, 1, 4, f) == 4;
}

int pt_write_png(const char *path, int w, int h, const unsigned char *rgb) {
	if(!crc_ready) crc_init();
	size_t row = 1 + 3 * (size_t)w, raw = row * (size_t)h;
	size_t nblocks = (raw + 65534) / 65535;
	if(nblocks == 0) nblocks = 1;
	size_t zlen = 2 + raw + 5 * nblocks + 4;
	if(zlen > 0x7FFFFFFFu) return 0;
	unsigned char *z = (unsigned char *)malloc(zlen), *filt = (unsigned char *)malloc(raw ? raw : 1);
	if(!z || !filt) { free(z); free(filt); return 0; }
	for(int y = 0; y < h; y++) {
		filt[row * (size_t)y] = 0; /* filter type None */
		memcpy(filt + row * (size_t)y + 1, rgb + 3 * (size_t)w * (size_t)y, 3 * (size_t)w);
	}
	size_t o = 0;
	z[o++] = 0x78; z[o++] = 0x01;
	uint32_t a = 1, b = 0;
	size_t pos = 0;
	for(size_t k = 0; k < nblocks; k++) {
		size_t n = raw - pos < 65535 ? raw - pos : 65535;
		z[o++] = (k + 1 == nblocks) ? 1 : 0;
		z[o++] = n & 0xFF; z[o++] = n >> 8; z[o++] = ~n & 0xFF; z[o++] = (~n >> 8) & 0xFF;
		memcpy(z + o, filt + pos, n);
		for(size_t i = 0; i < n; i++) { a = (a + filt[pos + i]) % 65521; b = (b + a) % 65521; }
		o += n; pos += n;
	}
	put_be32(z + o, (b << 16) | a); o += 4;
	FILE *f = fopen(path, "wb");
	int ok = f != NULL;
	if(ok) {
		static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
		unsigned char ihdr[13];
		put_be32(ihdr, (uint32_t)w); put_be32(ihdr + 4, (uint32_t)h);
		ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0; /* 8-bit RGB */
		ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr, 13) && write_chunk(f, "IDAT", z, (uint32_t)o) &&
		     write_chunk(f, "IEND", NULL, 0);
		ok = (fclose(f) == 0) && ok;
	}
	free(z); free(filt);
	return ok;
}

int pt_write_bmp(const char *path, int w, int h, const unsigned char *rgb) {
	int pad = (4 - (3 * w) % 4) % 4;
	uint32_t datasz = (uint32_t)((3 * w + pad) * h), filesz = 54 + datasz;
	unsigned char hd[54];
	memset(hd, 0, sizeof hd);
	hd[0] = 'B'; hd[1] = 'M';
	hd[2] = filesz; hd[3] = filesz >> 8; hd[4] = filesz >> 16; hd[5] = filesz >> 24;
	hd[10] = 54; hd[14] = 40;
	hd[18] = w; hd[19] = w >> 8; hd[20] = w >> 16; hd[21] = w >> 24;
	hd[22] = h; hd[23] = h >> 8; hd[24] = h >> 16; hd[25] = h >> 24;
	hd[26] = 1; hd[28] = 24;
	FILE *f = fopen(path, "wb");
	if(!f) return 0;
	int ok = fwrite(hd, 1, 54, f) == 54;
	static const unsigned char zero[3] = {0, 0, 0};
	for(int y = h - 1; y >= 0 && ok; y--) { /* bottom-up, BGR */
		for(int x = 0; x < w && ok; x++) {
			const unsigned char *p = rgb + 3 * ((size_t)y * w + x);
			unsigned char bgr[3] = {p[2], p[1], p[0]};
			ok = fwrite(bgr, 1, 3, f) == 3;
		}
		if(ok && pad) ok = fwrite(zero, 1, (size_t)pad, f) == (size_t)pad;
	}
	return (fclose(f) == 0) && ok;
}

int pt_write_tga(const char *path, int w, int h, const unsigned char *rgb) {
	unsigned char hd[18];
	memset(hd, 0, sizeof hd);
	hd[2] = 2; /* uncompressed true colour */
	hd[12] = w; hd[13] = w >> 8; hd[14] = h; hd[15] = h >> 8; hd[16] = 24;
	FILE *f = fopen(path, "wb");
	if(!f) return 0;
	int ok = fwrite(hd, 1, 18, f) == 18;
	for(int y = h - 1; y >= 0 && ok; y--) /* origin bottom-left, BGR */
		for(int x = 0; x < w && ok; x++) {
			const unsigned char *p = rgb + 3 * ((size_t)y * w + x);
			unsigned char bgr[3] = {p[2], p[1], p[0]};
			ok = fwrite(bgr, 1, 3, f) == 3;
		}
	return (fclose(f) == 0) && ok;
}
