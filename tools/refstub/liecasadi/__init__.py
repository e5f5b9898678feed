class SE3: pass
class SO3: pass
class Quaternion: pass
