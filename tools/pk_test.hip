#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_add(v2f a, v2f b){ v2f d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b){ v2f d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f add_mj(v2f a, v2f b){ v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f sub_mj(v2f a, v2f b){ v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f cmul(v2f a, v2f w){ v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t)); return d; }
__device__ __forceinline__ v2f cmul_s(v2f a, v2f w){ v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[0,0,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t)); return d; }
__global__ void k(const v2f* a, const v2f* b, v2f* o){
  int i = threadIdx.x;
  v2f x=a[i], y=b[i];
  v2f W = {0.92387953f, -0.38268343f};
  o[i*6+0]=pk_add(x,y); o[i*6+1]=pk_sub(x,y); o[i*6+2]=add_mj(x,y); o[i*6+3]=sub_mj(x,y); o[i*6+4]=cmul(x,y); o[i*6+5]=cmul_s(x,W);
}
int main(){
  v2f *a,*b,*o; hipMalloc(&a,64*8); hipMalloc(&b,64*8); hipMalloc(&o,64*48);
  v2f ha[64], hb[64], ho[64*6];
  for(int i=0;i<64;i++){ha[i]={1.0f+i,2.0f}; hb[i]={3.0f,-0.5f*i};}
  hipMemcpy(a,ha,sizeof ha,hipMemcpyHostToDevice); hipMemcpy(b,hb,sizeof hb,hipMemcpyHostToDevice);
  k<<<1,64>>>(a,b,o); hipMemcpy(ho,o,sizeof ho,hipMemcpyDeviceToHost);
  int bad=0;
  for(int i=0;i<64;i++){ float ax=ha[i].x,ay=ha[i].y,bx=hb[i].x,by=hb[i].y;
    float e[6][2]={{ax+bx,ay+by},{ax-bx,ay-by},{ax+by,ay-bx},{ax-by,ay+bx},{ax*bx-ay*by,ax*by+ay*bx},{ax*0.92387953f+ay*0.38268343f, -ax*0.38268343f+ay*0.92387953f}};
    for(int j=0;j<6;j++){ if(fabsf(ho[i*6+j].x-e[j][0])>1e-4f*fabsf(e[j][0])+1e-5f||fabsf(ho[i*6+j].y-e[j][1])>1e-4f*fabsf(e[j][1])+1e-5f){ if(bad<10)printf("bad i=%d j=%d got (%g,%g) exp (%g,%g)\n",i,j,ho[i*6+j].x,ho[i*6+j].y,e[j][0],e[j][1]); bad++; } } }
  printf("bad=%d\n",bad); return bad!=0;
}
