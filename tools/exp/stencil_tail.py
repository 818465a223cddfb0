"""The instrumental stage as a compact stencil (Next 1): the exact circular kernel g = irfft(taper) of the reference's
Gaussian taper exp(-2 pi^2 sigma^2 ss^2) (Payne/utils/smoothing.py:598-601) on N = 4096 points, the l1 mass of its tails beyond W taps
and the error a truncated stencil makes on a white 0.03-rms spectrum -- for the widths C2's prior produces (0.35 .. 2.5 resampled
pixels; 1.5 at the truth).  CPU only.  Result (NOTES R4.1): the taper is not small at the Nyquist frequency for these widths, the kernel
carries an alternating 1/j^2 tail, and a stencil exact to 1e-8 exists only above ~1.9 pixels (11 % of the prior, none of the posterior)."""
import numpy as np
N=4096
def kern(sig_px):
    k=np.arange(N//2+1)
    ss=k/N
    taper=np.exp(-2*np.pi**2*sig_px**2*ss**2)
    g=np.fft.irfft(taper,N)   # circular kernel, g[j] j=0..N-1
    return g, taper
rng=np.random.default_rng(0)
white=0.03*rng.normal(size=N)
for sig in [0.35,1.0,1.2,1.4,1.5,1.6,1.8,2.0,2.5]:
    g,t=kern(sig)
    row=[]
    for W in [int(np.ceil(6.5*sig))+1,33,64,128]:
        gt=g.copy(); idx=np.arange(N); d=np.minimum(idx,N-idx)
        tail=np.where(d>W,g,0.0)
        l1=np.abs(tail).sum()
        full=np.fft.irfft(np.fft.rfft(white)*t,N)
        tr=np.fft.irfft(np.fft.rfft(white)*np.fft.rfft(np.where(d<=W,g,0.0)),N)
        row.append((W,l1,np.abs(full-tr).max()))
    print("sig %.2f T(1/2)=%.2e"%(sig,t[-1]), " ".join("W=%d l1=%.1e err=%.1e"%r for r in row))
