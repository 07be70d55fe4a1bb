// pk_post.hip -- the packed real-input post-processing + squared magnitude of the pair (k, M - k) written as eight v_pk_* with
// explicit op_sel / neg modifiers (stft_fft.inl fft_post_pair_sq), against the plain formula: do the modifiers mean what
// stft_fft.inl assumes?   hipcc --offload-arch=gfx950 -O3 pk_post.hip -o pk_post && ./pk_post
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf fft_post_pair_sq(cf zk, cf zm, cf w, cf eps2) {
  cf S, D, t, wb, X, Y, q, r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(S) : "v"(zk), "v"(zm));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(zk), "v"(zm));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(D), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(wb) : "v"(D), "v"(w), "v"(t));
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(X) : "v"(S), "v"(wb));
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(Y) : "v"(S), "v"(wb));
  asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(q) : "v"(Y), "v"(eps2));
  asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(X), "v"(q));
  return r;
}
// the complex pair itself (for the Complex / Phase epilogues): xk = (S.x + wb.y, S.y - wb.x), xm = (S.x - wb.y, -(S.y + wb.x))
__device__ __forceinline__ void fft_post_pair_c(cf zk, cf zm, cf w, cf &xk, cf &xm) {
  cf S, D, t, wb;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(S) : "v"(zk), "v"(zm));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(zk), "v"(zm));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(D), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(wb) : "v"(D), "v"(w), "v"(t));
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(xk) : "v"(S), "v"(wb));                    // (S.x + wb.y, S.y - wb.x)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(xm) : "v"(S), "v"(wb));       // (S.x - wb.y, -S.y - wb.x)
}
__global__ void k(const cf *a, const cf *b, const cf *w, cf *o, cf *ok, cf *om, float eps) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  o[i] = fft_post_pair_sq(a[i], b[i], w[i], cf{eps, eps});
  cf xk, xm;
  fft_post_pair_c(a[i], b[i], w[i], xk, xm);
  ok[i] = xk;
  om[i] = xm;
}
int main() {
  const int n = 4096;
  std::vector<cf> a(n), b(n), w(n), o(n), xk(n), xm(n);
  for (int i = 0; i < n; ++i) {
    a[i] = cf{(float)sin(i * 0.37) * 3, (float)cos(i * 0.11) * 2};
    b[i] = cf{(float)sin(i * 0.73 + 1), (float)cos(i * 0.51 + 2) * 5};
    w[i] = cf{(float)cos(i * 0.001), (float)-sin(i * 0.001)};
  }
  cf *da, *db, *dw, *dout, *dk, *dm;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dw, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&dk, n * 8); hipMalloc(&dm, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dw, dout, dk, dm, 0.25f);
  hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost); hipMemcpy(xk.data(), dk, n * 8, hipMemcpyDeviceToHost); hipMemcpy(xm.data(), dm, n * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) {
    const double Sx = (double)a[i].x + b[i].x, Sy = (double)a[i].y - b[i].y, Dx = (double)a[i].x - b[i].x, Dy = (double)a[i].y + b[i].y;
    const double wbx = Dx * w[i].x - Dy * w[i].y, wby = Dx * w[i].y + Dy * w[i].x;
    const double kx = Sx + wby, ky = Sy - wbx, mx = Sx - wby, my = -(Sy + wbx);
    const double e[6] = {o[i].x - (kx * kx + ky * ky + 0.25), o[i].y - (mx * mx + my * my + 0.25), xk[i].x - kx, xk[i].y - ky, xm[i].x - mx, xm[i].y - my};
    for (double v : e) worst = fmax(worst, fabs(v));
  }
  printf("largest deviation from the plain formula: %.3e (values up to ~100)\n", worst);
  return worst < 1e-3 ? 0 : 1;
}
