class MeshcatVisualizer: pass
