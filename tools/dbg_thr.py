import sys, os, subprocess, tempfile, pathlib, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from helpers import build_sandbox_driver, cornell
from lumenrenderer_amd.scenes import write_scene_file
tmp=pathlib.Path(tempfile.mkdtemp())
scene=str(tmp/'c.slm'); write_scene_file(cornell(), scene)
exe=build_sandbox_driver(tmp)
W,H=96,64
for env_extra in ({}, {"SANDBOX_THREADED":"1"}, {"SANDBOX_THREADED":"1","SANDBOX_MOVE":"1"}):
    out=str(tmp/'o.ppm')
    r=subprocess.run([exe,scene,str(W),str(H),"3","24",out],capture_output=True,text=True,env=dict(os.environ,**env_extra))
    px=np.frombuffer(open(out,'rb').read()[-W*H*3:],np.uint8)
    print(env_extra, r.returncode, r.stdout.strip()[-120:], r.stderr.strip()[-200:], 'max',px.max(),'nonzero',(px>0).mean(), 'mean', px.mean())
