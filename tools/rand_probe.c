/* Does initialising the HIP runtime disturb libc's rand() stream?  (debug probe) */
#include <stdio.h>
#include <stdlib.h>
#include "pt_hip.h"
int main(void) {
	int n = 0;
	printf("before: ");
	pt_device_count(&n);
	pt_ctx *c = NULL;
	int rc = pt_ctx_create(0, &c);
	printf("devices %d rc %d\nafter ctx: ", n, rc);
	for(int i = 0; i < 4; i++) printf("%d ", rand());
	pt_image_alloc(c, 64, 64);
	printf("\nafter alloc: ");
	for(int i = 0; i < 4; i++) printf("%d ", rand());
	printf("\n");
	pt_ctx_destroy(c);
	return 0;
}
