import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import modelgen
from flash_viterbi_amd import decoder
g = json.load(open(os.path.join(ROOT, "tests/golden/cfg2_K3965_T256.json")))
A, B, Pi, ob = modelgen.model32(g["spec"])
fv = decoder.FlashViterbi(0); fv.set_model(A, B, Pi)
fv.set_option(decoder.OPT_KERNEL, int(sys.argv[1]) if len(sys.argv) > 1 else 4)
for _ in range(3): fv.decode_full(ob, 8, 0)
