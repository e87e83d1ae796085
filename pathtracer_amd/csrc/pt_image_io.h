/* pt_image_io.h — fallback image writers (see pt_image_io.c).  Return 1 on success, 0 on failure. */
#ifndef PT_IMAGE_IO_H
#define PT_IMAGE_IO_H
int pt_write_png(const char *path, int w, int h, const unsigned char *rgb);
int pt_write_bmp(const char *path, int w, int h, const unsigned char *rgb);
int pt_write_tga(const char *path, int w, int h, const unsigned char *rgb);
#endif
