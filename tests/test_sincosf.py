"""raytracer_amd/csrc/rtx_sincosf.h (the device's sinf/cosf) against the host
glibc on a dense sample of the range the hot path uses; compiled for the host
with g++ (same source the kernels include)."""
import os
import subprocess

from conftest import ROOT

PROG = r'''
#include "rtx_sincosf.h"
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static uint32_t u32(float f){uint32_t u; memcpy(&u,&f,4); return u;}
int main(){
  unsigned long long n=0,bad=0;
  // every 7th float in [2^-20, 8) both signs, plus the tiny/zero/edge cases
  float lo=0x1p-20f; uint32_t a=u32(lo), b=u32(8.0f);
  for(uint32_t u=a; u<b; u+=7){ float f; memcpy(&f,&u,4);
    for(int s=0;s<2;s++){ float y=s?-f:f; n++;
      if(u32(rtx_sinf(y))!=u32(sinf(y))||u32(rtx_cosf(y))!=u32(cosf(y))) bad++; } }
  float edge[]={0.0f,-0.0f,1e-30f,-1e-30f,0x1p-13f,0x1.921FB6p-1f,0x1.921fb6p+0f,0x1.921fb6p+1f,0x1.921fb6p+2f,6.2831855f,100.0f,119.99f};
  for(unsigned i=0;i<sizeof(edge)/4;i++){ n++; if(u32(rtx_sinf(edge[i]))!=u32(sinf(edge[i]))||u32(rtx_cosf(edge[i]))!=u32(cosf(edge[i]))) bad++; }
  printf("%llu %llu\n", n, bad); return 0; }
'''


def test_device_sincosf_matches_host_libm(tmp_path):
    src = tmp_path / "sc.cpp"
    src.write_text(PROG)
    exe = tmp_path / "sc"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"), str(src), "-o", str(exe), "-lm"])
    n, bad = map(int, subprocess.check_output([str(exe)]).split())
    assert n > 50_000_000
    assert bad == 0, f"{bad} of {n} results differ from glibc"
