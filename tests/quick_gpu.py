import sys, json, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
import numpy as np, modelgen
from flash_viterbi_amd import decoder
g=json.load(open('tests/golden/cfg2_K3965_T256.json'))
A,B,Pi,ob=modelgen.model32(g['spec'])
fv=decoder.FlashViterbi(0)
t=time.time(); fv.set_model(A,B,Pi); print('set_model s',time.time()-t)
ref=g['runs'][0]
for kern in (1,2):
  for mode in (0,1):
    fv.set_option(decoder.OPT_KERNEL,kern)
    for rep in range(3):
        p,s,rc=fv.decode_full(ob,8,mode)
    st=fv.stats()
    print('kern',kern,'mode',mode,'match',p.tolist()==ref['path'], s, ref['score'], {k:st[k] for k in ('decode_ms','gpu_ms','top_pass_ms','step_launches','task_steps','refine_near','refine_rescan','passes')})
    fv.set_option(decoder.OPT_PROFILE,1); p,s,rc=fv.decode_full(ob,8,mode); st=fv.stats(); fv.set_option(decoder.OPT_PROFILE,0)
    print('   profiled: step_kernel_ms',st['step_kernel_ms'],'launches',st['step_launches'],'avg us',1e3*st['step_kernel_ms']/st['step_launches'], 'gpu_ms',st['gpu_ms'])
