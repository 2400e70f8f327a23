#include "shim.h"
#include "CUDAKernels/RandomUtilities.cuh"
#include "Shaders/CppCommon/MaterialStructs.h"
#include "CUDAKernels/disney.cuh"
int main(){ unsigned s = WangHash(1); printf("%u %.9g\n", WangHash(0), RandomFloat(s));
 MaterialData m(0.f); m.SetColor(make_float4(.7f,.6f,.5f,1)); m.SetRoughness(.5f); m.SetMetallic(0); m.SetLuminance(1); m.SetRefractiveIndex(1/1.5f);
 float pdf; float3 n=make_float3(0,1,0), t=make_float3(1,0,0);
 float3 b = EvaluateBSDF(m,n,t,normalize(make_float3(.3f,.8f,.1f)),normalize(make_float3(-.2f,.9f,.3f)),pdf);
 printf("%.9g %.9g %.9g %.9g\n", b.x,b.y,b.z,pdf); }
