import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from zarc_amd import Engine, _lib
eng=Engine(0); eng.set_parameter(_lib.P_CHECKSUM_FLAG,1); eng.set_parameter(_lib.P_COMPRESSION_LEVEL,9)
n,size=2048,4<<20
off=np.arange(n,dtype=np.uint64)*np.uint64(size); lens=np.full(n,size,dtype=np.uint64)
cap=int(eng.bound(size))*n
d_src=eng.malloc(n*size+_lib.PAD); d_dst=eng.malloc(cap+_lib.PAD); d_out=eng.malloc(n*size+_lib.PAD)
eng.corpus_fill(d_src,off,lens,first_index=0,kind=2)
for rep in range(2):
    t0=time.perf_counter(); doff,dlen,dig,st=eng.pack_device(d_src,off,lens,d_dst,cap); t1=time.perf_counter()
    dig2,st2=eng.unpack_device(d_dst,doff,dlen,d_out,off,lens,expect=dig); t2=time.perf_counter()
print("level 9, %d x 4 MiB (kind lz): pack %.2f GiB/s unpack %.2f GiB/s ratio %.4f ok %s match_ms %.1f" % (n, n*size/(t1-t0)/2**30, n*size/(t2-t1)/2**30, n*size/float(dlen.sum()), bool((st2==0).all() and (dig2==dig).all()), eng.kernel_ms(_lib.T_MATCH)))
