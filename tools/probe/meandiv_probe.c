// Does q' = fma(fma(-q, b, x), r, q), q = x * r, r = RN(1 / b), equal RN(x / b) for every float x and every integer
// divisor b (the running mean of src/cl/main.cl:29 divides by frame + 1)?  Exhaustive over the 2^23 significands of x
// (the quotient scales with x's exponent away from under/overflow) for b = lo .. hi.
//   gcc -O2 -mfma -fopenmp -o /tmp/meandiv_probe tools/probe/meandiv_probe.c -lm && /tmp/meandiv_probe 1 4096
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
	const long lo = argc > 1 ? atol(argv[1]) : 1, hi = argc > 2 ? atol(argv[2]) : 4096, stride = argc > 3 ? atol(argv[3]) : 1;
	long bad = 0, badq = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad, badq)
	for(long bi = lo; bi <= hi; bi += stride) {
		const float b = (float)bi, r = 1.0f / b;
		for(uint32_t m = 0; m < (1u << 23); m++) {
			const uint32_t u = 0x3f800000u | m;
			float x;
			memcpy(&x, &u, 4);
			const float want = x / b;
			const float q = x * r;
			const float q2 = fmaf(fmaf(-q, b, x), r, q);
			if(q2 != want) {
				if(bad < 5) printf("mismatch b=%ld x=%a want=%a got=%a\n", bi, x, want, q2);
				bad++;
			}
			if(q != want) badq++;
		}
	}
	printf("b=%ld..%ld step %ld: %ld mismatches (the bare product differs %ld times)\n", lo, hi, stride, bad, badq);
	return bad != 0;
}
