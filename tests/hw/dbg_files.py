import sys, json, traceback
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import bench_configs as B
import oracle_lib as O
from rust_snappy_amd import raw, batch
dev=torch.device('cuda',0)
blob=dict(O.corpus_round())['zflat00_html']
for opts in ({}, {"lane_table_spread":0}):
    ctx=raw.Context(0)
    for k,v in opts.items(): ctx.set_option(k,v)
    gib=2.0
    reps=int(gib*2**30/len(blob)); stride=(len(blob)+15)//16*16
    one=np.zeros(stride,dtype=np.uint8); one[:len(blob)]=np.frombuffer(blob,dtype=np.uint8)
    data=torch.from_numpy(one).to(dev).repeat(reps)
    offs=np.arange(reps,dtype=np.int64)*stride; lens=np.full(reps,len(blob),dtype=np.int64)
    src=batch.StreamBatch(data,offs,lens)
    cap=raw.max_compress_len(len(blob))
    comp=batch.StreamBatch.empty(np.full(reps,cap,dtype=np.int64),dev)
    clens=torch.zeros(reps,dtype=torch.int64,device=dev)
    errs=torch.zeros(32*reps,dtype=torch.uint8,device=dev)
    raw.compress_batch(ctx,src.d_ptrs,src.d_lens,comp.d_ptrs,comp.d_lens,clens,errs,host_in_lens=src.h_lens)
    ctx.synchronize()
    want=O.compress(blob)
    cl=clens.cpu().numpy()
    bad_len=int((cl!=len(want)).sum())
    kinds=np.frombuffer(errs.cpu().numpy().tobytes(),dtype='<i4').reshape(reps,8)[:,0]
    w=torch.from_numpy(np.frombuffer(want,dtype=np.uint8).copy()).to(dev)
    rows=comp.data[:reps*int(comp.offsets[1])].view(reps,int(comp.offsets[1]))[:, :len(want)]
    badrows=(rows!=w[None,:]).any(dim=1)
    nb=int(badrows.sum())
    first=torch.nonzero(badrows)[:8].flatten().tolist()
    print(opts, "kernel",ctx.last_kernel(),"streams",reps,"bad lengths",bad_len,"errors",int((kinds!=0).sum()),"bad rows",nb,"first",first, ctx.table_probe_log()[:120], flush=True)
    if nb:
        i=first[0]
        r=rows[i].cpu().numpy(); ww=np.frombuffer(want,dtype=np.uint8)
        d=np.nonzero(r!=ww)[0]
        print("  stream",i,"first diff at",int(d[0]),"of",len(want),"ndiff",len(d), "len",int(cl[i]))
    ctx.close(); del data,comp; torch.cuda.empty_cache()
