import sys,os,torch
sys.path.insert(0,"ebfi-be_amd")
from ebfi_amd.engine import DEFAULT_MODEL_ARGS, Engine, synthetic_batch
from ebfi_amd import f16scale
eng = Engine(DEFAULT_MODEL_ARGS, device="cuda", precision="bf16x3", graph=False, seed=9, lr=1e-4)
for it in range(2):
    eng.train_step(*synthetic_batch(2, 128, 128, device="cuda", seed=500 + it, on_device=True))
batch = synthetic_batch(2, 128, 128, device="cuda", seed=502, on_device=True)
book = eng.book
snaps = []
orig_finish = book.finish
def finish():
    snaps.append(book.slots.clone())        # slots right before the finish launch: scale used in this pass, amax recorded in it
    orig_finish()
book.finish = finish
def gn(tag):
    eng.bucket.zero(); eng._fwd_bwd(*batch); torch.cuda.synchronize()
    f = eng.bucket.gather(); print(os.environ.get("TAG"), tag, "grad norm", float(f.double().norm()), "finite", bool(torch.isfinite(f).all()), flush=True)
gn("default"); gn("default")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    gn("side")
torch.cuda.synchronize()
sys.exit(0)
n = len(book.index)
inv = {v: k for k, v in book.index.items()}
S = f16scale.SLOT_STRIDE
a, b = snaps[1][: n * S].view(n, S).cpu(), snaps[2][: n * S].view(n, S).cpu()
print("slots whose recorded amax differs between the 2nd default pass and the side pass (same data, same weights):")
cnt = 0
for i in range(n):
    if a[i, 32] != b[i, 32] or a[i, 0] != b[i, 0]:
        cnt += 1
        if cnt <= 40:
            k = inv[i]
            print("  slot %3d %-4s scale %.3g -> %.3g   amax %.4g -> %.4g   floor %.3g -> %.3g" % (i, k[1] if isinstance(k, tuple) else k, a[i, 0], b[i, 0], a[i, 32], b[i, 32], a[i, 1], b[i, 1]))
print("differing:", cnt, "of", n)
