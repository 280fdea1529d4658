"""Debug aid: decode one small block with a chosen kernel variant and print the per-segment records."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zpaqsharp_amd as z
from tests import util
model = sys.argv[1] if len(sys.argv) > 1 else "mid"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
kern = int(sys.argv[3]) if len(sys.argv) > 3 else 8
data = util.text(n, seed=5)
s = util.block(model, data)
sc = z.scan(s)
a = np.frombuffer(s + b"\0" * 16, np.uint8)
d_in = torch.from_numpy(a.copy()).cuda()
d_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
with z.Context(0) as ctx:
    rc, res = ctx.decode_blocks_device(d_in.data_ptr(), len(s), sc, d_out.data_ptr(), [0], [n + 64], h_in=a, raise_on_error=False, kernel=kern)
    torch.cuda.synchronize()
    r = res[0]
    got = bytes(d_out.cpu().numpy()[:n])
    print("rc", rc, "status", r.status, "pp_state", hex(r.pp_state), "out_len", r.out_len, "in_used", hex(r.in_used), "out_off", hex(r.out_off), "of", sc.segments[0].data_len,
          "match" if got == data else "first diff at %d" % next((i for i in range(n) if got[i] != data[i]), -1))
