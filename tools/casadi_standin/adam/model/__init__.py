class Model: pass
