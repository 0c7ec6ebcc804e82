import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
pt = importlib.import_module("metal-pathtracer-arm64_amd")
bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")
host = pt.HostScene.load("scenes/cornell_mesh.scene", "scenes")
s = host.settings_for(width=1920, height=1080, max_depth=8, seed=1337)
scene = pt.DeviceScene(host.desc, 0, keepalive=host)
rows = bands.max_band_count(1080, 8) * 16
out = torch.zeros((rows, 1920, 3), dtype=torch.float32, device="cuda")
scene.render_device(s, 256, out.data_ptr(), 0, 0, 8, want_stats=False)
os.environ["PTR_TRACE_ITERATIONS"] = "1"
st = scene.render_device(s, 256, out.data_ptr(), 0, 0, 8, want_stats=True)
print("total", st.totalSeconds)
