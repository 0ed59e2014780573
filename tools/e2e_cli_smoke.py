import os, sys, subprocess, numpy as np, glob
from PIL import Image
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
tmp = "/tmp/e2ecli"; os.makedirs(tmp, exist_ok=True)
rng = np.random.default_rng(1)
vocab = ["<en_unk>", "a", "red", "green", "blue", "square", "is", "shown"]
colours = {"red": (220, 30, 30), "green": (30, 220, 30), "blue": (30, 30, 220)}
with open(tmp + "/sents.txt", "w") as f:
    for v in range(8):
        name = list(colours)[v % 3]
        os.makedirs(f"{tmp}/frames/vid{v}", exist_ok=True)
        for k in range(1, 12):
            img = np.clip(np.asarray(colours[name])[None, None, :] + rng.integers(-20, 20, (32, 32, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(f"{tmp}/frames/vid{v}/{k:06d}.jpg")
        f.write(f"vid{v}\ta {name} square is shown\nvid{v}\ta {name} square\n")
open(tmp + "/vocab.txt", "w").write("\n".join(vocab) + "\n")
open(tmp + "/attrs.txt", "w").write("red\ngreen\nblue\n")
env = dict(os.environ, PYTHONPATH=root)
base = [sys.executable, "-m", "s2vt_amd.train_e2e", "--train-sents", tmp + "/sents.txt", "--frames", tmp + "/frames", "--vocab", tmp + "/vocab.txt",
        "--epochs", "1", "--batch-size", "8", "--model-path", tmp + "/m"]
r = subprocess.run(base, env=env, cwd=tmp, capture_output=True, text=True); print("XE rc", r.returncode, r.stdout[-600:], r.stderr[-1500:])
ck = sorted(glob.glob(tmp + "/m/*model*"))
print(ck)
r = subprocess.run(base + ["--reinforce", "--samples", "2", "--attr-vocab", tmp + "/attrs.txt", "--test-sents", tmp + "/sents.txt"], env=env, cwd=tmp, capture_output=True, text=True)
print("RL rc", r.returncode, r.stdout[-1200:], r.stderr[-2500:])
