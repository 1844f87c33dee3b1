// Issue rate of packed against plain FP32 vector instructions on gfx950: N independent chains per lane, one full-machine launch
// each, instructions per cycle and SIMD from the kernel's duration and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/pk_rate scripts/probes/pk_rate.hip && scripts/probes/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, float seed)
{
	v2f a[CH];
	for (int k = 0; k < CH; ++k) a[k] = v2f{seed + k + threadIdx.x, seed - k};
	const v2f b = {seed * 0.5f, seed * 0.25f};
	for (int i = 0; i < ITER; ++i) {
#pragma unroll
		for (int k = 0; k < CH; ++k) {
			if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(b.x));
			if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(b.x));
			if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k].x) : "v"(b.x));
			if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
			if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
		}
	}
	float s = 0;
	for (int k = 0; k < CH; ++k) s += a[k].x + a[k].y;
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static double run(float* d, int blocks, const char* name, double clock_hz, int simds)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double waves = (double)blocks * 4, instr = waves * ITER * CH;
	const double per_simd_cycle = instr / (ms * 1e-3 * clock_hz * simds);
	printf("%-14s %8.3f ms  %.3f wave-instructions per cycle and SIMD  (%.2f cycles per instruction)\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle);
	return per_simd_cycle;
}

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, simds = cus * 4;
	const double clock_hz = p.clockRate * 1e3;
	const int blocks = cus * 8;   // 8 waves per SIMD
	float* d;
	hipMalloc(&d, (size_t)blocks * 256 * 4);
	printf("%s: %d CUs, %.0f MHz\n", p.gcnArchName, cus, clock_hz / 1e6);
	run<0>(d, blocks, "v_add_f32", clock_hz, simds);
	run<1>(d, blocks, "v_mul_f32", clock_hz, simds);
	run<2>(d, blocks, "v_fma_f32", clock_hz, simds);
	run<3>(d, blocks, "v_pk_add_f32", clock_hz, simds);
	run<4>(d, blocks, "v_pk_mul_f32", clock_hz, simds);
	run<5>(d, blocks, "v_pk_fma_f32", clock_hz, simds);
	hipFree(d);
	return 0;
}
