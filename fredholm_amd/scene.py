"""Host-side scene container: Python mirror of the reference's fredholm::Scene (fredholm/include/fredholm/scene.h:103-179,
fredholm/src/scene.cpp) -- .obj and .gltf ingestion into one set of flat arrays, the node hierarchy, and key-framed
animation of node transforms.  include/fredholm/scene.h is the C++ twin; both follow the same arithmetic (scene-graph math in
double, rounded to float once at the end) so that the two front ends hand bit-identical arrays to the library.

Reference behaviours kept on purpose (each is what the reference does, file:line):
  * .obj faces all get instance id 0 and every .obj shape appends one identity transform (scene.cpp:419-428);
  * glTF: one sub-mesh per node with a mesh, instance id = global sub-mesh index (scene.cpp:744-752); indices must be
    16-bit, POSITION/NORMAL float3, TEXCOORD_0 float2 with v -> 1 - v (scene.cpp:692-741); textures are all NONCOLOR
    and read from `image.uri` next to the file (scene.cpp:560-567); every material gets emission = 1 with the
    emissiveFactor as colour (scene.cpp:535-541); `clearcoatTexture` / `clearcoatRoughnessTexture` resolve to texture 0
    (GetNumberAsInt() of a JSON object, scene.cpp:520-531);
  * an animation drives the node named by its FIRST channel, which must be a root node of the scene
    (find_node_node drops the result of its recursion, scene.cpp:910-919), and REPLACES that node's transform by
    T * R * S of the interpolated keys (scene.cpp:862-893);
  * key interpolation mixes with h = t - input[idx0], NOT divided by the key interval (scene.h:164-178), with
    t = fmod(time, last key).
Deviation: Node::camera_id is never initialised in the reference (scene.cpp:669-672, undefined behaviour); here it is the
node's `camera` property or -1.
"""
import base64
import json
import math
import os

import numpy as np

from . import image_io
from .native import default_materials


def _identity():
    return [[1.0 if r == c else 0.0 for c in range(4)] for r in range(4)]


def _matmul(a, b):
    return [[a[r][0] * b[0][c] + a[r][1] * b[1][c] + a[r][2] * b[2][c] + a[r][3] * b[3][c] for c in range(4)] for r in range(4)]


def trs_matrix(t, q, s):
    """glm: translate(I, t) * mat4_cast(q) * scale(s); q = (w, x, y, z); row-major nested lists of Python floats (double)."""
    w, x, y, z = q
    r = [[1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - w * z), 2.0 * (x * z + w * y)],
         [2.0 * (x * y + w * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - w * x)],
         [2.0 * (x * z - w * y), 2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y)]]
    m = _identity()
    for i in range(3):
        for j in range(3):
            m[i][j] = r[i][j] * s[j]
        m[i][3] = t[i]
    return m


def affine_inverse(m):
    """inverse of [A t; 0 1] by cofactors, in double; the C++ facade uses the same expression order"""
    a, b, c = m[0][0], m[0][1], m[0][2]
    d, e, f = m[1][0], m[1][1], m[1][2]
    g, h, i = m[2][0], m[2][1], m[2][2]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    inv = 1.0 / det
    r = [[(e * i - f * h) * inv, (c * h - b * i) * inv, (b * f - c * e) * inv],
         [(f * g - d * i) * inv, (a * i - c * g) * inv, (c * d - a * f) * inv],
         [(d * h - e * g) * inv, (b * g - a * h) * inv, (a * e - b * d) * inv]]
    out = _identity()
    for k in range(3):
        for j in range(3):
            out[k][j] = r[k][j]
        out[k][3] = -(r[k][0] * m[0][3] + r[k][1] * m[1][3] + r[k][2] * m[2][3])
    return out


def _mix_vec(a, b, h):
    return [a[k] * (1.0 - h) + b[k] * h for k in range(3)]


def _mix_quat(x, y, a):
    """glm::mix(quat, quat, a): spherical interpolation without the shortest-path flip; linear when nearly parallel"""
    cos_theta = x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]
    if cos_theta > 1.0 - 1.1920928955078125e-07:
        return [x[k] * (1.0 - a) + y[k] * a for k in range(4)]
    angle = math.acos(cos_theta)
    s0, s1, sn = math.sin((1.0 - a) * angle), math.sin(a * angle), math.sin(angle)
    return [(s0 * x[k] + s1 * y[k]) / sn for k in range(4)]


def _interpolate(inputs, outputs, time, mix):
    """scene.h:164-178"""
    last = np.float32(inputs[-1])
    t = np.float32(math.fmod(float(np.float32(time)), float(last))) if last != 0 else np.float32(0.0)
    idx1 = int(np.searchsorted(np.asarray(inputs, dtype=np.float32), t, side="left"))
    idx0 = max(idx1 - 1, 0)
    idx1 = min(idx1, len(outputs) - 1)
    h = float(np.float32(t - np.float32(inputs[idx0])))
    return mix(outputs[idx0], outputs[idx1], h)


class Node:
    def __init__(self):
        self.idx = -1
        self.children = []
        self.transform = _identity()
        self.camera_id = -1
        self.submesh_id = -1


class Animation:
    def __init__(self):
        self.node = None
        self.translation_input, self.translation_output = [], []
        self.rotation_input, self.rotation_output = [], []  # quaternions as (w, x, y, z)
        self.scale_input, self.scale_output = [], []


_COMPONENT = {5120: (np.int8, 1), 5121: (np.uint8, 1), 5122: (np.int16, 2), 5123: (np.uint16, 2), 5125: (np.uint32, 4), 5126: (np.float32, 4)}
_TYPE_COUNT = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class Scene:
    def __init__(self):
        self.clear()

    def clear(self):
        self.m_has_camera_transform = False
        self.m_camera_transform = _identity()
        self.m_vertices = np.zeros((0, 3), np.float32)
        self.m_indices = np.zeros((0, 3), np.uint32)
        self.m_texcoords = np.zeros((0, 2), np.float32)
        self.m_normals = np.zeros((0, 3), np.float32)
        self.m_material_ids = np.zeros(0, np.uint32)
        self.m_materials = default_materials(0)
        self.m_textures = []           # {"rgba8": uint8[h, w, 4], "srgb": bool}
        self.m_submesh_offsets = []
        self.m_submesh_n_faces = []
        self.m_instance_ids = np.zeros(0, np.uint32)
        self.m_transforms = []         # 4x4 row-major nested lists (double)
        self.m_nodes = []
        self.m_animations = []

    def is_valid(self):
        return len(self.m_vertices) > 0 and len(self.m_indices) > 0 and len(self.m_normals) == len(self.m_vertices)

    # ------------------------------------------------------------------ scene.cpp:69-117
    def load_model(self, filepath, clear=True):
        if clear:
            self.clear()
        ext = os.path.splitext(str(filepath))[1]
        if ext == ".obj":
            self.load_obj(str(filepath))
        elif ext == ".gltf":
            self.load_gltf(str(filepath))
        else:
            raise ValueError(f"failed to load {filepath}: unsupported extension")

    def _append(self, vertices, normals, texcoords, indices, material_ids, instance_id):
        base = len(self.m_vertices)
        self.m_vertices = np.concatenate([self.m_vertices, np.asarray(vertices, np.float32).reshape(-1, 3)])
        self.m_normals = np.concatenate([self.m_normals, np.asarray(normals, np.float32).reshape(-1, 3)])
        self.m_texcoords = np.concatenate([self.m_texcoords, np.asarray(texcoords, np.float32).reshape(-1, 2)])
        idx = np.asarray(indices, np.uint32).reshape(-1, 3) + np.uint32(base)
        self.m_indices = np.concatenate([self.m_indices, idx])
        self.m_material_ids = np.concatenate([self.m_material_ids, np.asarray(material_ids, np.uint32)])
        self.m_instance_ids = np.concatenate([self.m_instance_ids, np.full(len(idx), instance_id, np.uint32)])
        return len(idx)

    def load_obj(self, filepath):
        from . import scenes
        d = scenes.load_obj(filepath)
        mat_base, tex_base = len(self.m_materials), len(self.m_textures)
        mats = d["materials"].copy()
        for name in mats.dtype.names:
            if name.endswith("_texture_id"):
                mats[name] = np.where(mats[name] >= 0, mats[name] + tex_base, mats[name])
        self.m_materials = np.concatenate([self.m_materials, mats])
        self.m_textures += d.get("textures") or []
        prev = len(self.m_indices)
        self._append(d["vertices"], d["normals"], d["texcoords"], d["indices"], d["material_ids"] + np.uint32(mat_base), 0)  # scene.cpp:424-428: instance id 0
        self.m_submesh_offsets.append(prev)
        self.m_submesh_n_faces.append(len(self.m_indices) - prev)
        self.m_transforms.append(_identity())

    # ------------------------------------------------------------------ scene.cpp:445-860
    def load_gltf(self, filepath):
        try:
            self._load_gltf(filepath)
        except ValueError:
            raise
        except Exception as e:  # missing keys, indices outside their arrays, accessors outside their buffers ...
            raise ValueError(f"failed to load {filepath}: inconsistent glTF ({type(e).__name__}: {e})") from e

    def _load_gltf(self, filepath):
        try:
            model = json.load(open(filepath, "r"))
        except (OSError, ValueError) as e:
            raise ValueError(f"failed to load {filepath}") from e
        folder = os.path.dirname(filepath)
        buffers = []
        for b in model.get("buffers", []):
            uri = b.get("uri", "")
            if uri.startswith("data:"):
                buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                try:
                    buffers.append(open(os.path.join(folder, uri), "rb").read())
                except OSError as e:
                    raise ValueError(f"failed to load {filepath}: buffer {uri}") from e

        def get_buffer(accessor_id):  # scene.cpp:921-933: (bytes from the accessor's start, stride, count)
            acc = model["accessors"][accessor_id]
            if "bufferView" not in acc:
                raise ValueError("accessor without bufferView")
            view = model["bufferViews"][acc["bufferView"]]
            comp = _COMPONENT[acc["componentType"]][1] * _TYPE_COUNT[acc["type"]]
            stride = view.get("byteStride", 0) or comp
            start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
            return buffers[view["buffer"]], start, stride, acc["count"]

        def read(accessor_id, dtype, per, stride_want, what):
            data, start, stride, count = get_buffer(accessor_id)
            if stride != stride_want:
                raise ValueError(what)
            return np.frombuffer(data, dtype=dtype, count=count * per, offset=start).reshape(count, per) if per > 1 else np.frombuffer(data, dtype=dtype, count=count, offset=start), count

        mat_base, tex_base = len(self.m_materials), len(self.m_textures)
        gm = model.get("materials", [])
        mats = default_materials(len(gm))
        for i, material in enumerate(gm):
            pmr = material.get("pbrMetallicRoughness", {})
            m = mats[i]
            m["base_color"] = pmr.get("baseColorFactor", [1, 1, 1, 1])[:3]
            if "baseColorTexture" in pmr: m["base_color_texture_id"] = pmr["baseColorTexture"]["index"] + tex_base
            m["specular_roughness"] = pmr.get("roughnessFactor", 1.0)
            m["metalness"] = pmr.get("metallicFactor", 1.0)
            if "metallicRoughnessTexture" in pmr: m["metallic_roughness_texture_id"] = pmr["metallicRoughnessTexture"]["index"] + tex_base
            cc = material.get("extensions", {}).get("KHR_materials_clearcoat")
            if cc is not None:
                if "clearcoatFactor" in cc: m["coat"] = cc["clearcoatFactor"]
                if "clearcoatTexture" in cc: m["coat_texture_id"] = 0  # scene.cpp:521-523: GetNumberAsInt() of an object
                if "clearcoatRoughnessFactor" in cc: m["coat_roughness"] = cc["clearcoatRoughnessFactor"]
                if "clearcoatRoughnessTexture" in cc: m["coat_roughness_texture_id"] = 0
            m["emission"] = 1.0  # scene.cpp:535-541: tinygltf always yields a 3-vector
            m["emission_color"] = material.get("emissiveFactor", [0, 0, 0])
            if "emissiveTexture" in material: m["emission_texture_id"] = material["emissiveTexture"]["index"] + tex_base
            if "normalTexture" in material: m["normalmap_texture_id"] = material["normalTexture"]["index"] + tex_base
        self.m_materials = np.concatenate([self.m_materials, mats])
        for texture in model.get("textures", []):
            image = model["images"][texture["source"]]
            self.m_textures.append({"rgba8": image_io.load_texture(os.path.join(folder, image["uri"]), flip_vertically=True), "srgb": False})

        def load_node(node_idx):
            node = model["nodes"][node_idx]
            n = Node()
            n.idx = node_idx
            t = [float(v) for v in node.get("translation", [0, 0, 0])]
            r = node.get("rotation")
            q = [1.0, 0.0, 0.0, 0.0] if r is None else [float(r[3]), float(r[0]), float(r[1]), float(r[2])]
            s = [float(v) for v in node.get("scale", [1, 1, 1])]
            # the reference builds the matrix in float (glm); keys are rounded to float first, the products are done in double
            t, q, s = [float(np.float32(v)) for v in t], [float(np.float32(v)) for v in q], [float(np.float32(v)) for v in s]
            n.transform = trs_matrix(t, q, s)
            if "matrix" in node:  # column-major in the file
                mm = node["matrix"]
                n.transform = [[float(np.float32(mm[4 * c + r_])) for c in range(4)] for r_ in range(4)]
            n.camera_id = node.get("camera", -1)
            if "mesh" in node:
                mesh = model["meshes"][node["mesh"]]
                n.submesh_id = len(self.m_submesh_offsets)
                prev = len(self.m_indices)
                for prim in mesh["primitives"]:
                    idx, n_idx = read(prim["indices"], np.uint16, 1, 2, "indices stride is not ushort")
                    attrs = prim["attributes"]
                    pos, n_pos = read(attrs["POSITION"], np.float32, 3, 12, "positions stride is not float3")
                    nrm, _ = read(attrs["NORMAL"], np.float32, 3, 12, "normals stride is not float3")
                    uv, _ = read(attrs["TEXCOORD_0"], np.float32, 2, 8, "texcoord stride is not float2")
                    uv = np.stack([uv[:, 0], np.float32(1.0) - uv[:, 1]], axis=1)
                    tri = idx[: (n_idx // 3) * 3].astype(np.uint32).reshape(-1, 3)
                    material = np.uint32(prim.get("material", -1) & 0xFFFFFFFF) + np.uint32(mat_base if prim.get("material", -1) >= 0 else 0)
                    self._append(pos, nrm, uv, tri, np.full(len(tri), material, np.uint32), len(self.m_submesh_offsets))
                self.m_submesh_offsets.append(prev)
                self.m_submesh_n_faces.append(len(self.m_indices) - prev)
            for child in node.get("children", []):
                n.children.append(load_node(child))
            return n

        first_new_node = len(self.m_nodes)
        for node_idx in model["scenes"][0]["nodes"]:
            self.m_nodes.append(load_node(node_idx))
        while len(self.m_transforms) < len(self.m_submesh_offsets):
            self.m_transforms.append(_identity())
        self.update_transform()

        for animation in model.get("animations", []):
            anim = Animation()
            target = animation["channels"][0]["target"]["node"]
            anim.node = next((n for n in self.m_nodes[first_new_node:] if n.idx == target), None)  # root nodes only (scene.cpp:900-919)
            if anim.node is None:
                raise ValueError("invalid target node")
            for channel in animation["channels"]:
                sampler = animation["samplers"][channel["sampler"]]
                path = channel["target"]["path"]
                inp, n_in = read(sampler["input"], np.float32, 1, 4, "unsupported animation input")
                data, start, stride, n_out = get_buffer(sampler["output"])
                if n_in != n_out:
                    raise ValueError("animation input size is not equal to output size")
                per = {"translation": 3, "rotation": 4, "scale": 3}.get(path)
                if per is None:
                    continue
                if stride != 4 * per:
                    raise ValueError("invalid output stride")
                out = np.frombuffer(data, dtype=np.float32, count=n_out * per, offset=start).reshape(n_out, per)
                if path == "translation":
                    anim.translation_input += [float(v) for v in inp]
                    anim.translation_output += [[float(v) for v in o] for o in out]
                elif path == "rotation":
                    anim.rotation_input += [float(v) for v in inp]
                    anim.rotation_output += [[float(o[3]), float(o[0]), float(o[1]), float(o[2])] for o in out]
                else:
                    anim.scale_input += [float(v) for v in inp]
                    anim.scale_output += [[float(v) for v in o] for o in out]
            self.m_animations.append(anim)

    # ------------------------------------------------------------------ scene.cpp:836-898
    def update_transform(self):
        def visit(node, parent):
            m = _matmul(parent, node.transform)
            if node.camera_id != -1:
                self.m_has_camera_transform = True
                self.m_camera_transform = m
            if node.submesh_id != -1:
                self.m_transforms[node.submesh_id] = m
            for c in node.children:
                visit(c, m)
        for node in self.m_nodes:
            visit(node, _identity())

    def update_animation(self, time):
        for a in self.m_animations:
            t = _interpolate(a.translation_input, a.translation_output, time, _mix_vec) if a.translation_input else [0.0, 0.0, 0.0]
            q = _interpolate(a.rotation_input, a.rotation_output, time, _mix_quat) if a.rotation_input else [1.0, 0.0, 0.0, 0.0]
            s = _interpolate(a.scale_input, a.scale_output, time, _mix_vec) if a.scale_input else [1.0, 1.0, 1.0]
            a.node.transform = trs_matrix(t, q, s)
        self.update_transform()

    # ------------------------------------------------------------------ what Renderer::load_scene uploads (renderer.h:361-421)
    def transforms_3x4(self):
        o2w = np.asarray([[m[r][c] for r in range(3) for c in range(4)] for m in self.m_transforms], dtype=np.float32).reshape(-1, 12)
        w2o = np.asarray([[affine_inverse(m)[r][c] for r in range(3) for c in range(4)] for m in self.m_transforms], dtype=np.float32).reshape(-1, 12)
        return o2w, w2o

    def camera_transform_3x4(self):
        return np.asarray([[self.m_camera_transform[r][c] for c in range(4)] for r in range(3)], dtype=np.float32)

    def as_dict(self):
        if not self.is_valid():
            raise ValueError("invalid scene")
        o2w, w2o = self.transforms_3x4()
        d = {"vertices": self.m_vertices, "normals": self.m_normals, "texcoords": self.m_texcoords, "indices": self.m_indices, "material_ids": self.m_material_ids,
             "materials": self.m_materials, "instance_ids": self.m_instance_ids, "object_to_world": o2w, "world_to_object": w2o}
        if self.m_textures:
            d["textures"] = self.m_textures
        return d
