import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import afp_oracle as O
from audfprint_amd.batch import Extractor
ex=Extractor.get(0); ex.set_params()
clips=[O.synth_noise(i,30.0) for i in range(32)]
r=ex.extract(clips=clips, debug=True)
print('flags', np.bincount(r.unit_flags, minlength=8))
st=ex.debug(4,np.float64,(4,))
print(st[:3])
logS=ex.debug(0,np.float64,(256,))
print('min logS', logS.min(), 'logfloor', st[0,0])
