"""Write a scratch tree with the file names the reference drivers hard-code (raypng.c:28,75-81):
scenes/render.map + procedural PNG stand-ins for the assets.   python tools/make_assets.py DIR [skybox_width] [scene]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from example_gui_opencl_raytracer_amd import api, scene, textures
d = sys.argv[1]
skyw = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
which = sys.argv[3] if len(sys.argv) > 3 else "render_map"
for sub in ("scenes", "assets/bg", "out"):
    os.makedirs(os.path.join(d, sub), exist_ok=True)
sc = {"render_map": scene.render_map_scene, "c3": lambda: scene.dielectric_field_scene(8), "c4": lambda: scene.sphere_grid_scene(100, 100)}[which]()
sc.save(os.path.join(d, "scenes", "render.map"))
t = textures.texture_layers()
for i, n in enumerate(("cobblestone", "sand", "check", "grass")):
    api.write_png_rgba(os.path.join(d, "assets", n + ".png"), t[i])
api.write_png_rgba(os.path.join(d, "assets", "bg", "stormydays.png"), textures.skybox_cross(skyw)[0])
print("wrote", d)
