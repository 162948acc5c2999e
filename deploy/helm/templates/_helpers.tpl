{{/* Shared naming, label and resource helpers of the chart. */}}

{{- define "stack.engineLabels" -}}
{{- with .Values.servingEngineSpec.labels }}
{{- toYaml . }}
{{- end }}
{{- end }}

{{- define "stack.routerLabels" -}}
{{- with .Values.routerSpec.labels }}
{{- toYaml . }}
{{- end }}
{{- end }}

{{- define "stack.cacheLabels" -}}
{{- if and .Values.cacheserverSpec .Values.cacheserverSpec.labels }}
{{- toYaml .Values.cacheserverSpec.labels }}
{{- else }}
app: {{ .Release.Name }}-cache-server
{{- end }}
{{- end }}

{{/* "k1=v1,k2=v2" selector string the router's pod watch uses (keys sorted). */}}
{{- define "stack.engineSelector" -}}
{{- $pairs := list -}}
{{- range $k := (keys .Values.servingEngineSpec.labels | sortAlpha) -}}
{{- $pairs = append $pairs (printf "%s=%s" $k (index $.Values.servingEngineSpec.labels $k)) -}}
{{- end -}}
{{- join "," $pairs -}}
{{- end }}

{{/* Rollout strategy: the spec's own, or surge a full new set before taking the old one down. */}}
{{- define "stack.strategy" -}}
{{- if .strategy }}
{{- toYaml .strategy }}
{{- else }}
rollingUpdate:
  maxSurge: 100%
  maxUnavailable: 0
{{- end }}
{{- end }}

{{/* Container resources of one modelSpec.  GPUs are requested through the AMD device plugin
     resource unless the spec names another one. */}}
{{- define "stack.engineResources" -}}
{{- $ms := . -}}
{{- $gpu := default "amd.com/gpu" $ms.requestGPUType -}}
{{- $vendor := (splitList "/" $gpu | first) -}}
requests:
  memory: {{ required "modelSpec.requestMemory must be set" $ms.requestMemory | quote }}
  cpu: {{ required "modelSpec.requestCPU must be set" $ms.requestCPU | quote }}
  {{- if gt (int (default 0 $ms.requestGPU)) 0 }}
  {{ $gpu }}: {{ $ms.requestGPU | quote }}
  {{- end }}
  {{- if $ms.requestGPUMem }}
  {{ $vendor }}/gpumem: {{ $ms.requestGPUMem | quote }}
  {{- end }}
  {{- if $ms.requestGPUMemPercentage }}
  {{ $vendor }}/gpumem-percentage: {{ $ms.requestGPUMemPercentage | quote }}
  {{- end }}
  {{- if $ms.requestGPUCores }}
  {{ $vendor }}/gpucores: {{ $ms.requestGPUCores | quote }}
  {{- end }}
{{- if or $ms.limitMemory $ms.limitCPU (gt (int (default 0 $ms.requestGPU)) 0) $ms.limitGPUMem $ms.limitGPUMemPercentage $ms.limitGPUCores }}
limits:
  {{- if $ms.limitMemory }}
  memory: {{ $ms.limitMemory | quote }}
  {{- end }}
  {{- if $ms.limitCPU }}
  cpu: {{ $ms.limitCPU | quote }}
  {{- end }}
  {{- if gt (int (default 0 $ms.requestGPU)) 0 }}
  {{ $gpu }}: {{ $ms.requestGPU | quote }}
  {{- end }}
  {{- if $ms.limitGPUMem }}
  {{ $vendor }}/gpumem: {{ $ms.limitGPUMem | quote }}
  {{- end }}
  {{- if $ms.limitGPUMemPercentage }}
  {{ $vendor }}/gpumem-percentage: {{ $ms.limitGPUMemPercentage | quote }}
  {{- end }}
  {{- if $ms.limitGPUCores }}
  {{ $vendor }}/gpucores: {{ $ms.limitGPUCores | quote }}
  {{- end }}
{{- end }}
{{- end }}

{{/* env entry body for a value that is either a literal (kept in <release>-secrets under
     .key) or a {secretName, secretKey} reference. */}}
{{- define "stack.secretEnv" -}}
valueFrom:
  secretKeyRef:
    {{- if kindIs "string" .value }}
    name: {{ .release }}-secrets
    key: {{ .key }}
    {{- else }}
    name: {{ .value.secretName }}
    key: {{ .value.secretKey }}
    {{- end }}
{{- end }}

{{/* KV role of an engine from its transfer role. */}}
{{- define "stack.kvRole" -}}
{{- if eq (toString .nixlRole) "sender" -}}kv_producer
{{- else if eq (toString .nixlRole) "receiver" -}}kv_consumer
{{- else -}}kv_both
{{- end -}}
{{- end }}

{{/* lm://host:port of the release's cache server. */}}
{{- define "stack.cacheServerUrl" -}}
lm://{{ .Release.Name }}-cache-server-service:{{ default 81 .Values.cacheserverSpec.servicePort }}
{{- end }}
