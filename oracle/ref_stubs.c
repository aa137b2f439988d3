/*
 * ref_stubs.c -- TEST INFRASTRUCTURE.  Satisfies the four GPU-backend externs that the
 * reference's run.c references on Linux (reference src/run.c:22-25) so that the untouched
 * reference CLI links as a CPU-only binary (oracle/_ref/run_cpu, always run with CALM_CPU=1).
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

struct Transformer;

static void no_gpu(const char* what) {
	fprintf(stderr, "%s: this oracle build has no GPU backend; run with CALM_CPU=1\n", what);
	abort();
}

void* upload_cuda(void* host, size_t size) {
	(void)host, (void)size;
	no_gpu("upload_cuda");
	return NULL;
}

void prepare_cuda(struct Transformer* transformer) {
	(void)transformer;
	no_gpu("prepare_cuda");
}

float* forward_cuda(struct Transformer* transformer, int token, int pos, unsigned flags) {
	(void)transformer, (void)token, (void)pos, (void)flags;
	no_gpu("forward_cuda");
	return NULL;
}

void perf_cuda(void) {
}
