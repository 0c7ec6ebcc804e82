"""Edge sizes on one MI355X: a 4096-spp frame of config 2 against its 256-spp image, a 1x1 frame, a 7x5 frame at depth 1."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("metal-pathtracer-arm64_amd")
host = pt.HostScene.load(os.path.join(ROOT, "scenes", "cornell_mesh.scene"), os.path.join(ROOT, "scenes"))
dev = pt.DeviceScene(host.desc, 0, keepalive=host)
s = host.settings_for(width=1920, height=1080, max_depth=8, seed=1337)
a, sa = dev.render_image(s, 256)
t0 = time.time()
b, sb = dev.render_image(s, 4096)
print("4096 spp: %.2f s, %.0f Msamples/s" % (sb.totalSeconds, 1920 * 1080 * 4096 / sb.totalSeconds / 1e6))
lum = np.array([0.2126, 0.7152, 0.0722])
print("mean lum ratio 4096/256:", float((b @ lum).mean() / (a @ lum).mean()), "rmse", float(np.sqrt(np.mean((a - b) ** 2))), "finite", bool(np.isfinite(b).all()))
# 4K, depth 12, many spp on the large scene is covered by full_configs; here: a 1-spp render and a 1x1 render
s1 = host.settings_for(width=1, height=1, max_depth=8, seed=1)
c, _ = dev.render_image(s1, 1024)
print("1x1x1024:", c.reshape(-1))
s2 = host.settings_for(width=7, height=5, max_depth=1, seed=1)
print("7x5 depth 1:", dev.render_image(s2, 3)[0].shape)
