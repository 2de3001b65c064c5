import sys, collections, traceback
sys.path.insert(0, '/root/repo')
import torch, bench
args = bench.parse_args([])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.VocoderWorkload(args, 0, 1, dev); wl.model.finalize(dev); wl.prepare(0, args.batch)
wl.eager_step(); torch.cuda.synchronize()
cnt = collections.Counter()
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'megatts2_hierspeechpp_amd' in fr.filename:
            return f"{fr.filename.split('/')[-1]}:{fr.lineno} {fr.line[:70]}"
    return "?"
oc, ocl, ocp, oto = torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_, torch.Tensor.to
def c(self, *a, **k):
    if self.is_cuda and not self.is_contiguous(): cnt[("contiguous", where())] += 1
    return oc(self, *a, **k)
def cl(self, *a, **k):
    if self.is_cuda: cnt[("clone", where())] += 1
    return ocl(self, *a, **k)
def cp(self, *a, **k):
    if self.is_cuda: cnt[("copy_", where())] += 1
    return ocp(self, *a, **k)
def to(self, *a, **k):
    r = oto(self, *a, **k)
    if self.is_cuda and r is not self: cnt[("to", where())] += 1
    return r
torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_, torch.Tensor.to = c, cl, cp, to
wl.eager_step(); torch.cuda.synchronize()
for k, v in cnt.most_common(30): print(v, k)
