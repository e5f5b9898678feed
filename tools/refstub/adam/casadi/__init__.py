class KinDynComputations: pass
