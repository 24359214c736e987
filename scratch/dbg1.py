import sys; sys.path.insert(0,'.')
import numpy as np, torch
from map_neo_amd.minhash import Engine, pack_texts, as_u32
from oracle import ref_c as RC
e = Engine(0)
for n in (77, 78, 100, 141):
    doc = ("abcdefghijklmnopqrstuvwxyz "*40)[:n]
    buf, offs = pack_texts([doc])
    t = e.to_device(buf)
    got = e.shingle_hashes(t, n, 0, n)
    exp = RC.shingle_hashes(doc.encode())
    bad = np.nonzero(got!=exp)[0]
    print(n, bad.tolist())
    for i in bad[:3]:
        # does got[i] equal any expected?
        print('  ', i, hex(got[i]), hex(exp[i]), np.nonzero(exp==got[i])[0].tolist())
