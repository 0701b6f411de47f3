"""Developer diagnosis: where the device packer and the host packer disagree."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

def run(alphabet, cfg, scale, label, sync_before=False, stream=None, n=5000):
    sdb = synth.make_config_db(cfg, scale=scale)
    seq, off = synth.make_reads(alphabet, n, 120, seed=12, amb_rate=0.004, bad_rate=0.01, var_len=60)
    db = ra.PhyloKmerDB.from_synth(sdb); pp = ra.PlacementProcess(db)
    packed, lens, flags = pp.pack_reads_host(seq, off)
    outs = []
    for trial in range(3):
        s_t = torch.from_numpy(seq).cuda(); o_t = torch.from_numpy(off.astype(np.int64)).cuda()
        if sync_before:
            torch.cuda.synchronize()
        if stream is not None:
            with torch.cuda.stream(stream):
                dpk, dl, df = pp.pack_reads(s_t, o_t, int(lens.max()))
            stream.synchronize()
        else:
            dpk, dl, df = pp.pack_reads(s_t, o_t, int(lens.max()))
        d = dpk.cpu().numpy().view(np.uint32)
        outs.append(d)
        idx = np.argwhere(d != packed)
        print(label, "trial", trial, "diff words", len(idx), "rows", len(np.unique(idx[:, 0])) if len(idx) else 0)
    db.close()

run(20, "C4", 0.2, "AA null-stream")
run(20, "C4", 0.2, "AA sync-before", sync_before=True)
run(20, "C4", 0.2, "AA side-stream", stream=torch.cuda.Stream())
run(4, "C1", 1.0, "DNA null-stream")
run(20, "C4", 0.2, "AA null-stream n=50000", n=50000)
