/* Empty stand-in.  The reference's src/c/main.c:7 includes <CL/cl.h> without using anything from
 * it; when that file is compiled unchanged against this backend (pathtracer_amd/csrc/Makefile,
 * target `pathtracer`) the include resolves here, so no OpenCL SDK is needed. */
