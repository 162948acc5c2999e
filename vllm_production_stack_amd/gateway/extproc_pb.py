"""Protobuf messages of Envoy's external-processing protocol, built at import time from
descriptors written out by hand (there is no protoc and no Envoy API package on the build hosts).

Only the fields an endpoint picker reads or writes are declared; protobuf keeps fields it does not
know as unknown fields, so messages from a newer Envoy still parse.  Field numbers and names follow
the public API:

  envoy/service/ext_proc/v3/external_processor.proto   ProcessingRequest / ProcessingResponse,
      HttpHeaders, HttpBody, HttpTrailers, HeadersResponse, BodyResponse, TrailersResponse,
      CommonResponse, ImmediateResponse, GrpcStatus, HeaderMutation, BodyMutation
  envoy/config/core/v3/base.proto                      HeaderMap, HeaderValue, HeaderValueOption
  envoy/type/v3/http_status.proto                      HttpStatus
  grpc/health/v1/health.proto                          HealthCheckRequest / HealthCheckResponse

The service is ``envoy.service.ext_proc.v3.ExternalProcessor`` with the one bidirectional method
``Process``.  The reference reaches the same protocol through the upstream endpoint-picker binary it
patches its pickers into (/root/reference/src/gateway_inference_extension/scheduler.patch:13-77,
/root/reference/src/gateway_inference_extension/Dockerfile).
"""

from __future__ import annotations

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, struct_pb2, wrappers_pb2

F = descriptor_pb2.FieldDescriptorProto
_OPT, _REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED


def _msg(file, name, fields, oneofs=(), enums=()):
    m = file.message_type.add()
    m.name = name
    for o in oneofs:
        m.oneof_decl.add().name = o
    for ename, values in enums:
        e = m.enum_type.add()
        e.name = ename
        for vname, num in values:
            v = e.value.add()
            v.name, v.number = vname, num
    for spec in fields:
        fname, num, ftype = spec[0], spec[1], spec[2]
        kw = spec[3] if len(spec) > 3 else {}
        f = m.field.add()
        f.name, f.number, f.type = fname, num, ftype
        f.label = kw.get("label", _OPT)
        if "type_name" in kw:
            f.type_name = kw["type_name"]
        if "oneof" in kw:
            f.oneof_index = list(oneofs).index(kw["oneof"])
    return m


def _build():
    pool = descriptor_pool.Default()
    # dependencies that ship with protobuf are in the default pool already
    assert struct_pb2.DESCRIPTOR and wrappers_pb2.DESCRIPTOR

    core = descriptor_pb2.FileDescriptorProto()
    core.name = "vpsa/envoy/config/core/v3/base_subset.proto"
    core.package = "envoy.config.core.v3"
    core.syntax = "proto3"
    core.dependency.append("google/protobuf/wrappers.proto")
    _msg(core, "HeaderValue", [("key", 1, F.TYPE_STRING), ("value", 2, F.TYPE_STRING), ("raw_value", 3, F.TYPE_BYTES)])
    _msg(core, "HeaderMap", [("headers", 1, F.TYPE_MESSAGE, {"label": _REP, "type_name": ".envoy.config.core.v3.HeaderValue"})])
    _msg(core, "HeaderValueOption",
         [("header", 1, F.TYPE_MESSAGE, {"type_name": ".envoy.config.core.v3.HeaderValue"}),
          ("append", 2, F.TYPE_MESSAGE, {"type_name": ".google.protobuf.BoolValue"}),
          ("append_action", 3, F.TYPE_ENUM, {"type_name": ".envoy.config.core.v3.HeaderValueOption.HeaderAppendAction"}),
          ("keep_empty_value", 4, F.TYPE_BOOL)],
         enums=[("HeaderAppendAction", [("APPEND_IF_EXISTS_OR_ADD", 0), ("ADD_IF_ABSENT", 1),
                                        ("OVERWRITE_IF_EXISTS_OR_ADD", 2), ("OVERWRITE_IF_EXISTS", 3)])])

    typ = descriptor_pb2.FileDescriptorProto()
    typ.name = "vpsa/envoy/type/v3/http_status_subset.proto"
    typ.package = "envoy.type.v3"
    typ.syntax = "proto3"
    _msg(typ, "HttpStatus", [("code", 1, F.TYPE_INT32)])  # an enum upstream; same wire form

    ep = descriptor_pb2.FileDescriptorProto()
    ep.name = "vpsa/envoy/service/ext_proc/v3/external_processor_subset.proto"
    ep.package = "envoy.service.ext_proc.v3"
    ep.syntax = "proto3"
    ep.dependency.extend([core.name, typ.name, "google/protobuf/struct.proto"])
    P = ".envoy.service.ext_proc.v3."
    _msg(ep, "HttpHeaders", [("headers", 1, F.TYPE_MESSAGE, {"type_name": ".envoy.config.core.v3.HeaderMap"}),
                             ("end_of_stream", 3, F.TYPE_BOOL)])
    _msg(ep, "HttpBody", [("body", 1, F.TYPE_BYTES), ("end_of_stream", 2, F.TYPE_BOOL)])
    _msg(ep, "HttpTrailers", [("trailers", 1, F.TYPE_MESSAGE, {"type_name": ".envoy.config.core.v3.HeaderMap"})])
    _msg(ep, "ProcessingRequest",
         [("request_headers", 2, F.TYPE_MESSAGE, {"type_name": P + "HttpHeaders", "oneof": "request"}),
          ("response_headers", 3, F.TYPE_MESSAGE, {"type_name": P + "HttpHeaders", "oneof": "request"}),
          ("request_body", 4, F.TYPE_MESSAGE, {"type_name": P + "HttpBody", "oneof": "request"}),
          ("response_body", 5, F.TYPE_MESSAGE, {"type_name": P + "HttpBody", "oneof": "request"}),
          ("request_trailers", 6, F.TYPE_MESSAGE, {"type_name": P + "HttpTrailers", "oneof": "request"}),
          ("response_trailers", 7, F.TYPE_MESSAGE, {"type_name": P + "HttpTrailers", "oneof": "request"}),
          ("observability_mode", 10, F.TYPE_BOOL)],
         oneofs=["request"])
    _msg(ep, "HeaderMutation",
         [("set_headers", 1, F.TYPE_MESSAGE, {"label": _REP, "type_name": ".envoy.config.core.v3.HeaderValueOption"}),
          ("remove_headers", 2, F.TYPE_STRING, {"label": _REP})])
    _msg(ep, "BodyMutation", [("body", 1, F.TYPE_BYTES, {"oneof": "mutation"}),
                              ("clear_body", 2, F.TYPE_BOOL, {"oneof": "mutation"})], oneofs=["mutation"])
    _msg(ep, "CommonResponse",
         [("status", 1, F.TYPE_ENUM, {"type_name": P + "CommonResponse.ResponseStatus"}),
          ("header_mutation", 2, F.TYPE_MESSAGE, {"type_name": P + "HeaderMutation"}),
          ("body_mutation", 3, F.TYPE_MESSAGE, {"type_name": P + "BodyMutation"}),
          ("trailers", 4, F.TYPE_MESSAGE, {"type_name": ".envoy.config.core.v3.HeaderMap"}),
          ("clear_route_cache", 5, F.TYPE_BOOL)],
         enums=[("ResponseStatus", [("CONTINUE", 0), ("CONTINUE_AND_REPLACE", 1)])])
    _msg(ep, "HeadersResponse", [("response", 1, F.TYPE_MESSAGE, {"type_name": P + "CommonResponse"})])
    _msg(ep, "BodyResponse", [("response", 1, F.TYPE_MESSAGE, {"type_name": P + "CommonResponse"})])
    _msg(ep, "TrailersResponse", [("header_mutation", 1, F.TYPE_MESSAGE, {"type_name": P + "HeaderMutation"})])
    _msg(ep, "GrpcStatus", [("status", 1, F.TYPE_UINT32)])
    _msg(ep, "ImmediateResponse",
         [("status", 1, F.TYPE_MESSAGE, {"type_name": ".envoy.type.v3.HttpStatus"}),
          ("headers", 2, F.TYPE_MESSAGE, {"type_name": P + "HeaderMutation"}),
          ("body", 3, F.TYPE_BYTES),
          ("grpc_status", 4, F.TYPE_MESSAGE, {"type_name": P + "GrpcStatus"}),
          ("details", 5, F.TYPE_STRING)])
    _msg(ep, "ProcessingResponse",
         [("request_headers", 1, F.TYPE_MESSAGE, {"type_name": P + "HeadersResponse", "oneof": "response"}),
          ("response_headers", 2, F.TYPE_MESSAGE, {"type_name": P + "HeadersResponse", "oneof": "response"}),
          ("request_body", 3, F.TYPE_MESSAGE, {"type_name": P + "BodyResponse", "oneof": "response"}),
          ("response_body", 4, F.TYPE_MESSAGE, {"type_name": P + "BodyResponse", "oneof": "response"}),
          ("request_trailers", 5, F.TYPE_MESSAGE, {"type_name": P + "TrailersResponse", "oneof": "response"}),
          ("response_trailers", 6, F.TYPE_MESSAGE, {"type_name": P + "TrailersResponse", "oneof": "response"}),
          ("immediate_response", 7, F.TYPE_MESSAGE, {"type_name": P + "ImmediateResponse", "oneof": "response"}),
          ("dynamic_metadata", 8, F.TYPE_MESSAGE, {"type_name": ".google.protobuf.Struct"})],
         oneofs=["response"])

    health = descriptor_pb2.FileDescriptorProto()
    health.name = "vpsa/grpc/health/v1/health_subset.proto"
    health.package = "grpc.health.v1"
    health.syntax = "proto3"
    _msg(health, "HealthCheckRequest", [("service", 1, F.TYPE_STRING)])
    _msg(health, "HealthCheckResponse",
         [("status", 1, F.TYPE_ENUM, {"type_name": ".grpc.health.v1.HealthCheckResponse.ServingStatus"})],
         enums=[("ServingStatus", [("UNKNOWN", 0), ("SERVING", 1), ("NOT_SERVING", 2), ("SERVICE_UNKNOWN", 3)])])

    out = {}
    for fdp in (core, typ, ep, health):
        fd = pool.Add(fdp) if hasattr(pool, "Add") else pool.AddSerializedFile(fdp.SerializeToString())
        if fd is None:
            fd = pool.FindFileByName(fdp.name)
        for name in fd.message_types_by_name:
            out[name] = message_factory.GetMessageClass(fd.message_types_by_name[name])
    return out


_classes = _build()
HeaderValue = _classes["HeaderValue"]
HeaderMap = _classes["HeaderMap"]
HeaderValueOption = _classes["HeaderValueOption"]
HttpStatus = _classes["HttpStatus"]
HttpHeaders = _classes["HttpHeaders"]
HttpBody = _classes["HttpBody"]
HttpTrailers = _classes["HttpTrailers"]
ProcessingRequest = _classes["ProcessingRequest"]
ProcessingResponse = _classes["ProcessingResponse"]
HeaderMutation = _classes["HeaderMutation"]
BodyMutation = _classes["BodyMutation"]
CommonResponse = _classes["CommonResponse"]
HeadersResponse = _classes["HeadersResponse"]
BodyResponse = _classes["BodyResponse"]
TrailersResponse = _classes["TrailersResponse"]
ImmediateResponse = _classes["ImmediateResponse"]
GrpcStatus = _classes["GrpcStatus"]
HealthCheckRequest = _classes["HealthCheckRequest"]
HealthCheckResponse = _classes["HealthCheckResponse"]

SERVICE = "envoy.service.ext_proc.v3.ExternalProcessor"
METHOD = f"/{SERVICE}/Process"
HEALTH_SERVICE = "grpc.health.v1.Health"
OVERWRITE_IF_EXISTS_OR_ADD = 2
