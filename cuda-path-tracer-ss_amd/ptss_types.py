"""ctypes mirrors of include/ptss_types.h (field order = the reference's RenderStructs.h / Primitives.h)."""
import ctypes as C


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Quat(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class UChar4(C.Structure):
    _fields_ = [("x", C.c_ubyte), ("y", C.c_ubyte), ("z", C.c_ubyte), ("w", C.c_ubyte)]


class Sphere(C.Structure):
    _fields_ = [("position", Vec3), ("radius", C.c_float), ("materialIdx", C.c_int)]


class Triangle(C.Structure):
    _fields_ = [("vertex0", Vec3), ("vertex1", Vec3), ("vertex2", Vec3),
                ("normal0", Vec3), ("normal1", Vec3), ("normal2", Vec3), ("materialIdx", C.c_int)]


class Material(C.Structure):
    _fields_ = [("diffuseColor", Vec3), ("specularColor", Vec3), ("absorption", Vec3), ("emmitance", Vec3),
                ("specularExponent", C.c_float), ("indexOfRefraction", C.c_float), ("diffAvg", C.c_float),
                ("specAvg", C.c_float), ("refrAvg", C.c_float), ("roughness", C.c_float), ("flags", C.c_char)]


class PointLight(C.Structure):
    _fields_ = [("position", Vec3), ("power", Vec3)]


class AreaLight(C.Structure):
    _fields_ = [("power", Vec3), ("area", C.c_float), ("triangleIdx", C.c_int), ("numTriangles", C.c_size_t)]


class Camera(C.Structure):
    _fields_ = [("rotation", Quat), ("position", Vec3), ("zNear", C.c_float), ("zFar", C.c_float),
                ("fieldOfView", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("numSpheres", C.c_size_t),
                ("triangles", C.POINTER(Triangle)), ("numTriangles", C.c_size_t),
                ("materials", C.POINTER(Material)), ("numMaterials", C.c_size_t),
                ("pointLights", C.POINTER(PointLight)), ("numPointLights", C.c_size_t),
                ("areaLights", C.POINTER(AreaLight)), ("numAreaLights", C.c_size_t),
                ("defaultColor", Vec3)]


assert C.sizeof(Sphere) == 20 and C.sizeof(Triangle) == 76 and C.sizeof(Material) == 76
assert C.sizeof(PointLight) == 24 and C.sizeof(AreaLight) == 32 and C.sizeof(Camera) == 40


def struct_to_dict(s):
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        if isinstance(v, C.Structure):
            out[name] = struct_to_dict(v)
        elif isinstance(v, bytes):
            out[name] = v[0] if v else 0
        else:
            out[name] = v
    return out
