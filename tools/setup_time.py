import sys, time
sys.path.insert(0,'cuda-path-tracer-ss_amd')
import ptss
sc=ptss.Scene('stress')
r=ptss.Renderer(sc,64,64); r.close()
t=time.perf_counter(); r=ptss.Renderer(sc,64,64); print('create stress %.1f ms'%((time.perf_counter()-t)*1e3)); r.close()
sc=ptss.Scene('mixed')
t=time.perf_counter(); r=ptss.Renderer(sc,64,64); print('create mixed %.1f ms'%((time.perf_counter()-t)*1e3)); r.close()
