"""The window kernel with its scheduler forced (span_schedule 2) on a small and
a mid-size batch, against the oracle; prints as it goes (a hang shows where)."""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import oracle_lib as O
from rust_snappy_amd import raw, batch
rnd=[d for _,d in O.corpus_round()]
want=[O.compress(d) for d in rnd]
for rounds in (1, 4, 30, 88):
    for mode in (2, 0):
        ctx=raw.Context(0)
        ctx.set_option("compress_mode",0); ctx.set_option("small_batch_kernel",0)
        ctx.set_option("span_schedule",mode)
        src=batch.StreamBatch.from_bytes(rnd*rounds)
        print("rounds",rounds,"mode",mode,"...",flush=True)
        t0=time.perf_counter()
        dst,lens,errs=batch.compress(ctx,src)
        ctx.synchronize()
        t1=time.perf_counter()
        dst,lens,errs=batch.compress(ctx,src)
        ctx.synchronize()
        t2=time.perf_counter()
        bad=[i for i in range(len(rnd)*rounds) if dst.stream_bytes(i,lens[i])!=want[i%12]]
        print("   ",ctx.last_kernel(),"first %.2f ms, again %.2f ms, mismatches %d"%((t1-t0)*1e3,(t2-t1)*1e3,len(bad)),flush=True)
        ctx.close()
