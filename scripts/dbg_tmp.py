import os, sys
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (32, 32, 32)
def worker(rank, world, port, real):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import param_fill
    from sgnn_amd import synth
    from sgnn_amd.model import GenModel
    from sgnn_amd.train import GraphStep, to_device
    dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
    model = param_fill(GenModel(8, DIMS, 1, 16, 16, 4, True, True, 1, 1), 5).train().to(dev)
    batches = [to_device(synth.make_batch(2, DIMS, cfg=7, first_block=10 * it + 2 * rank, occupancy=0.08), dev) for it in range(2)]
    box = {}
    def sync(flat):
        torch.cuda.synchronize()
        g = flat[:box['s'].opt.numel]
        bad = torch.nonzero(~torch.isfinite(g)).view(-1)
        msg = 'local grads finite'
        if bad.numel():
            i0 = int(bad[0]); o = 0
            for (segname, ps) in box['s'].opt.segments:
                for p_ in ps:
                    if o <= i0 < o + p_.numel():
                        nm = [n for n, q in model.named_parameters() if q is p_][0]
                        msg = 'LOCAL bad grad idx %d in %s (%d bad)' % (i0, nm, bad.numel())
                    o += p_.numel()
        print('   rank %d pre-reduce: %s' % (rank, msg), flush=True)
        w = model.encoder.encode_dense0[0].weight
        if w.grad is not None and not torch.isfinite(w.grad).all():
            b = ~torch.isfinite(w.grad)
            print('   rank %d grad shape %s stride %s bad per cout %s per cin %s per tap(first 16) %s' % (rank, tuple(w.grad.shape), w.grad.stride(),
                  b.sum((1,2,3,4)).tolist(), b.sum((0,2,3,4)).tolist(), b.reshape(24,16,64).sum((0,1))[:16].tolist()), flush=True)
            bn = model.encoder.encode_dense0[1]
            print('   rank %d bn grads finite %s %s; other dense weights finite %s' % (rank, bool(torch.isfinite(bn.weight.grad).all()), bool(torch.isfinite(bn.bias.grad).all()),
                  [bool(torch.isfinite(m[0].weight.grad).all()) for m in (model.encoder.encode_dense1, model.encoder.bottleneck_dense2, model.encoder.decode_dense3, model.encoder.decode_dense4, model.encoder.final)]), flush=True)
        if real:
            dist.all_reduce(flat)
    step = GraphStep(model, lr=1e-3, headroom=1.6, grad_sync=sync, world_size=world if real else 1)
    box['s'] = step
    lw = np.ones(5, dtype=np.float32)
    for it in range(4):
        loss = step(batches[it % 2], lw)
        torch.cuda.synchronize()
        print('rank %d real %s it %d stage %d loss %.4f params finite %s' % (rank, real, it, step.stage, float(loss), bool(torch.isfinite(step.opt.flat_p).all())), flush=True)
        dist.barrier()
    dist.destroy_process_group()
if __name__ == '__main__':
    for real in (True,):
        mp.spawn(worker, args=(2, 38123 + int(real), real), nprocs=2, join=True)
