import numpy as np, scipy.fft as sf, time
rng=np.random.default_rng(0)
def run(C,k,L,d=1,N=128):
    x=rng.standard_normal((C,L)).astype(np.float32)
    w=(rng.standard_normal((C,C,k))/np.sqrt(C*k)).astype(np.float32)
    # direct float64 reference (same padding: zero pad (k-1)*d/2)
    pad=(k-1)*d//2
    xp=np.pad(x.astype(np.float64),((0,0),(pad,pad)))
    ref=np.zeros((C,L))
    for j in range(k):
        ref+=w[:,:,j].astype(np.float64)@xp[:,j*d:j*d+L]
    # direct fp32 (k-ordered fma chain emulated by float32 matmul accumulate)
    y32=np.zeros((C,L),np.float32)
    xp32=np.pad(x,((0,0),(pad,pad)))
    for j in range(k):
        y32+=w[:,:,j]@xp32[:,j*d:j*d+L]
    # overlap-save FFT in float32 per polyphase component
    hop=N-(k-1)
    yf=np.zeros((C,L),np.float32)
    W=sf.rfft(np.pad(w,((0,0),(0,0),(0,N-k))),axis=2).astype(np.complex64)   # [co,ci,bins]
    for p in range(d):
        xs=xp32[:,p::d]            # subsampled, includes padding
        Lp=xs.shape[1]-(k-1)       # outputs of this phase
        nseg=(Lp+hop-1)//hop
        xs=np.pad(xs,((0,0),(0,nseg*hop+k-1-xs.shape[1])))
        segs=np.stack([xs[:,s*hop:s*hop+N] for s in range(nseg)],1)      # [ci,seg,N]
        X=sf.rfft(segs,axis=2)                                           # complex64
        Y=np.einsum('oib,isb->osb',np.conj(W),X).astype(np.complex64)
        y=sf.irfft(Y,n=N,axis=2).astype(np.float32)[:,:,:hop]           # valid part: N-(k-1)
        # correlation vs convolution: conv1d is correlation: y[t]=sum_j w[j] x[t+j]; use flipped filter
        yph=y.reshape(C,-1)[:,:Lp]
        # outputs of phase p correspond to t = p + d*i - ... handled below
        yf_phase=yph
        idx=np.arange(Lp)*d+p
        m=idx<L+0
        # position mapping: xp index t' = t (output t uses xp[t + j d]); phase p of xp -> outputs t with t%d==p
        yf[:,idx[m]]=yf_phase[:,m][:, :m.sum()]
    return ref,y32,yf
for C,k,d in [(128,11,1),(512,11,1),(256,7,3),(128,11,5)]:
    L=2000
    # correlation: flip filter for FFT path
    x=None
    ref,y32,yf=run(C,k,L,d)
    print(C,k,d,'max|ref|',np.abs(ref).max(),'direct32 err',np.abs(y32-ref).max(),'fft32 err',np.abs(yf-ref).max())
